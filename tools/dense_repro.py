import numpy as np, sys, os
sys.path.insert(0, '.')
from orb_ygz_slam_amd import Extractor
from oracle import oracle_py as oracle
for (w, h, nl, sf, nf, per) in [(556, 515, 6, 2.0, 2469, 16), (538, 632, 3, 2.0, 1578, 16), (400, 300, 4, 1.2, 1000, 16), (400, 300, 4, 1.2, 1000, 20), (752, 480, 8, 1.2, 1000, 16)]:
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.clip((((xx % per) - per // 2) ** 2 + ((yy % per) - per // 2) ** 2) * (250.0 / (2 * (per // 2) ** 2)), 0, 255).astype(np.uint8)
    print("cfg", w, h, nl, sf, nf, per, flush=True)
    ex = Extractor(nf, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=1)
    k, d = ex.extract(img)
    ok, od = oracle.Extractor(nf, sf, nl, 20, 7).extract(img)
    print("  gpu", len(k), "oracle", len(ok), "equal", len(k) == len(ok) and bool((k == ok).all() and (d == od).all()), flush=True)
