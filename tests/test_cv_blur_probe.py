"""The shell's detector of the linked OpenCV's GaussianBlur generation (orb_ygz_slam_amd/csrc/host/cv_blur_probe.h): its three integer models equal
the oracle's three definitions on the probe image, the three differ from each other there (one pixel -- the exact tie -- between the two legacy forms),
every generation's blur is recognised and anything else is rejected.  (The call of the real cv::GaussianBlur lives in host/ORBextractor.cc and needs
an OpenCV this image does not have; everything around it is exercised here.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("bp") / "libblur_probe.so")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-shared", "-fPIC", os.path.join(ROOT, "tests", "cpp", "blur_probe_capi.cc"), "-o", so])
    L = C.CDLL(so)
    w, h = L.bp_width(), L.bp_height()
    img = np.zeros((h, w), np.uint8)
    L.bp_image(img.ctypes.data_as(C.c_void_p))
    return L, img


def test_models_equal_the_oracles_three_generations(oracle, probe):
    L, img = probe
    outs = []
    for mode in (0, 1, 2):
        want = np.zeros_like(img)
        L.bp_expected(mode, want.ctypes.data_as(C.c_void_p))
        with oracle.cv_mode(mode):
            got = oracle.blur(img)
        assert np.array_equal(want, got), mode
        assert L.bp_classify(np.ascontiguousarray(got).ctypes.data_as(C.c_void_p)) == mode
        outs.append(want)
    d01 = np.argwhere(outs[0] != outs[1])
    assert len(d01) >= 1 and (d01[:, 1] == 3).all()              # the legacy forms part on the tie column only (128 half-to-even, 129 half-up)
    assert outs[0][0, 3] == 128 and outs[1][0, 3] == 129
    assert (outs[2] != outs[0]).sum() > 20                        # the Q8.8 kernel differs all over


def test_anything_else_is_rejected(probe):
    L, img = probe
    want = np.zeros_like(img)
    L.bp_expected(0, want.ctypes.data_as(C.c_void_p))
    off = want.copy()
    off[5, 9] ^= 1
    assert L.bp_classify(off.ctypes.data_as(C.c_void_p)) == -1
    assert L.bp_classify(img.ctypes.data_as(C.c_void_p)) == -1   # the unblurred image


def test_the_cv_adapter_recognises_each_generation(tmp_path):
    """blur_probe_run_opencv -- what ygz::ORBextractor calls against a real OpenCV -- over the oracle-backed cv:: stand-in, whose GaussianBlur is switched
    through the three generations: the adapter must name each."""
    so = str(tmp_path / "libblur_probe_cv.so")
    shim = os.path.join(ROOT, "oracle", "ref_shim")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-shared", "-fPIC", "-ffp-contract=off", "-I" + shim, "-I" + os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "cpp", "blur_probe_cv_capi.cc"), os.path.join(shim, "mini_cv.cpp"), os.path.join(ROOT, "oracle", "oracle_cvprims.cpp"),
                           os.path.join(ROOT, "oracle", "oracle_fast10.cpp"), "-o", so])
    L = C.CDLL(so)
    for mode in (0, 1, 2):
        L.yo_set_cv_mode(mode)
        assert L.bp_run_opencv() == mode
    L.yo_set_cv_mode(0)
