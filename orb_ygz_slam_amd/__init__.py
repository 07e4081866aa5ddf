"""orb_ygz_slam_amd -- MI355X-native ORB / direct visual-SLAM front end (the per-frame hot path of
gaoxiang12/ORB-YGZ-SLAM) behind a C ABI (include/ygzf.h).

This Python package is plumbing only: a ctypes binding of lib/libygzf.so used by the tests and bench.py.  The product
is the shared library (hand-written HIP kernels + C++ host code).  There is NO CPU fallback: importing works without a
GPU (so that the CPU test tier can check the exported symbols), but creating a context raises without a HIP device.
"""
from .capi import Extractor, MultiGpu, YgzfError, load_library, KP_DTYPE, Camera, make_camera, EUROC  # noqa: F401
