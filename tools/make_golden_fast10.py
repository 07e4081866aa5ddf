#!/usr/bin/env python3
"""Generate tests/golden/fast10_test1.npz from the REFERENCE's libfast (oracle/_ref/libfast_ref.so, compiled from
/root/reference/Thirdparty/fast/src) run on the reference's own test image Thirdparty/fast/test/data/test1.png.

Pinned known answer: 167 corners at threshold 75 (Thirdparty/fast/test/test.cpp:52 "BENCHMARK version extracted 167
features").  Needs /root/reference + Pillow; the .npz it writes is committed so the GPU box (no reference) can use it.
"""
import os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O

img = np.array(Image.open("/root/reference/Thirdparty/fast/test/data/test1.png").convert("L"), np.uint8)
assert img.shape == (480, 752)
out = {"image": img}
for thr in (75, 20):
    xy, sc, nm = O.ref_fast10(img, thr, which=1)
    out["xy_%d" % thr], out["score_%d" % thr], out["nonmax_%d" % thr] = xy, sc, nm
    print(thr, len(xy), len(nm))
assert len(out["xy_75"]) == 167
np.savez_compressed(os.path.join(ROOT, "tests/golden/fast10_test1.npz"), **out)
