#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle diagnosis for one frame (run on the GPU box through gpurun)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as O
from orb_ygz_slam_amd import Extractor
from orb_ygz_slam_amd.synth import synth_frame

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
img = synth_frame(0, w, h)
ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
oex = O.Extractor(1000, 1.2, 8, 20, 7)
t0 = time.time(); ex.extract_batch_host(img[None]); ex.sync(); print("gpu extract", time.time() - t0)
pyr = oex.pyramid(img)
for l in range(8):
    g = ex.batch_fetch_level(0, l)
    print("pyr", l, g.shape, "diff px:", int((g != pyr[l]).sum()))
for l in range(8):
    xs, ys, sc = oex.cell_candidates(l)
    gx, gy, gs = ex.batch_fetch_candidates(0, l)
    same = len(gx) == len(xs) and (gx == xs).all() and (gy == ys).all() and (gs == sc).all()
    print("cand", l, len(xs), len(gx), "OK" if same else "DIFF")
    if not same:
        a = set(zip(xs.tolist(), ys.tolist(), sc.tolist())); b = set(zip(gx.tolist(), gy.tolist(), gs.tolist()))
        print("   only oracle:", sorted(a - b)[:10], " only gpu:", sorted(b - a)[:10])
ok, od = oex.extract(img)
for l in range(8):
    kl = oex.level_keypoints(l)
    gx, gy, gs = ex.batch_fetch_level_keypoints(0, l)
    same = len(gx) == len(kl) and (gx == kl["x"].astype(np.int32)).all() and (gy == kl["y"].astype(np.int32)).all()
    print("oct", l, len(kl), len(gx), "OK" if same else "DIFF")
    if not same:
        a = set(zip(kl["x"].astype(int).tolist(), kl["y"].astype(int).tolist())); b = set(zip(gx.tolist(), gy.tolist()))
        print("   set equal:", a == b, " only oracle:", sorted(a - b)[:6], " only gpu:", sorted(b - a)[:6])
k, d = ex.batch_fetch(0)
print("final", len(ok), len(k))
if len(k) == len(ok):
    for fld in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        print("  ", fld, "mismatch:", int((k[fld] != ok[fld]).sum()))
    print("   desc rows differing:", int((d != od).any(axis=1).sum()))
