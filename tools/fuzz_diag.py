#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle diagnosis of one fuzz case of tests/test_gpu_extract.py::test_extract_fuzz_sizes_and_configs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as O
from orb_ygz_slam_amd import Extractor
from orb_ygz_slam_amd.synth import synth_frame

seed = int(sys.argv[1])
rng = np.random.default_rng(100 + seed)
w, h = int(rng.integers(120, 900)), int(rng.integers(100, 700))
nl = int(rng.integers(1, 10))
sf = float(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0]))
nf = int(rng.integers(50, 3000))
ini = int(rng.integers(8, 40))
mn = int(rng.integers(2, ini + 1))
print("config", w, h, nl, sf, nf, ini, mn)
ex = Extractor(nf, sf, nl, ini, mn, max_width=w, max_height=h, max_batch=2)
oex = O.Extractor(nf, sf, nl, ini, mn)
imgs = np.stack([synth_frame(200 + seed, w, h), rng.integers(0, 256, (h, w), dtype=np.uint8)])
ex.extract_batch_host(imgs)
for f in range(2):
    ok, od = oex.extract(imgs[f])
    pyr = oex.pyramid(imgs[f])
    for l in range(nl):
        g = ex.batch_fetch_level(f, l)
        if (g != pyr[l]).any(): print("frame", f, "pyr", l, "diff px", int((g != pyr[l]).sum()))
        xs, ys, sc = oex.cell_candidates(l)
        gx, gy, gs = ex.batch_fetch_candidates(f, l)
        same = len(gx) == len(xs) and (gx == xs).all() and (gy == ys).all() and (gs == sc).all()
        kl = oex.level_keypoints(l)
        ox, oy, osc = ex.batch_fetch_level_keypoints(f, l)
        same2 = len(ox) == len(kl) and (ox == kl["x"].astype(np.int32)).all() and (oy == kl["y"].astype(np.int32)).all()
        print("frame", f, "level", l, "cand", len(xs), len(gx), "OK" if same else "DIFF", "| oct", len(kl), len(ox), "OK" if same2 else "DIFF")
        if not same:
            a = set(zip(xs.tolist(), ys.tolist(), sc.tolist())); b = set(zip(gx.tolist(), gy.tolist(), gs.tolist()))
            print("   only oracle:", sorted(a - b)[:8], " only gpu:", sorted(b - a)[:8])
        elif not same2:
            a = set(zip(kl["x"].astype(int).tolist(), kl["y"].astype(int).tolist())); b = set(zip(ox.tolist(), oy.tolist()))
            print("   set equal:", a == b, " only oracle:", sorted(a - b)[:6], " only gpu:", sorted(b - a)[:6])
    k, d = ex.batch_fetch(f)
    print("frame", f, "final", len(ok), len(k))
