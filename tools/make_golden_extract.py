#!/usr/bin/env python3
"""Golden vectors of the ORACLE's extractor + matcher on seeded synthetic frames -> tests/golden/extract_golden.npz.

The reference ships no vectors for this path (parity unpinned), so these pin OUR oracle's definition:
regressions of the oracle show up on CPU, and the GPU tier can compare the HIP path against committed bytes without
trusting a freshly built oracle.  Deterministic: numpy default_rng seeds, integer/float32 arithmetic only."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O
from orb_ygz_slam_amd.synth import synth_frame

CASES = [("vga_s0", 640, 480, 0, (1000, 1.2, 8, 20, 7)), ("vga_s1", 640, 480, 1, (1000, 1.2, 8, 20, 7)),
         ("euroc_s2", 752, 480, 2, (1000, 1.2, 8, 20, 7)), ("euroc_4lvl_s3", 752, 480, 3, (1000, 2.0, 4, 20, 7)),
         ("small_s4", 320, 240, 4, (500, 1.2, 8, 20, 7))]
MODES = (O.CV_MODE_LEGACY_SSE2, O.CV_MODE_LEGACY_INT, O.CV_MODE_CV4)   # the three GaussianBlur definitions (oracle_cvprims.cpp)
out = {}
for name, w, h, seed, cfg in CASES:
    img = synth_frame(seed, w, h)
    out[name + "_img_sha"] = np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8)
    for mode in MODES:
        with O.cv_mode(mode):
            k, d = O.Extractor(*cfg).extract(img)
        out["%s_m%d_sha" % (name, mode)] = np.frombuffer(hashlib.sha256(k.tobytes() + d.tobytes()).digest(), np.uint8)
        out[name + "_n"] = np.array([len(k)])    # keypoints do not depend on the blur mode
        if mode == O.CV_MODE_LEGACY_SSE2 and name in ("vga_s0", "small_s4"):   # full vectors for two frames (others: hashes only)
            out[name + "_kps"] = k
            out[name + "_desc"] = d
        print(name, mode, len(k), hashlib.sha256(k.tobytes() + d.tobytes()).hexdigest()[:16])
np.savez_compressed(os.path.join(ROOT, "tests/golden/extract_golden.npz"), **out)
