#!/usr/bin/env python3
"""tools/make_golden_frame_ref.py -- what THE REFERENCE'S OWN src/Frame.cc / src/MapPoint.cc return (oracle/_ref/libref_frame.so, libref_mappoint.so, built by
oracle/Makefile from the checkout) on the scenarios of tests/frame_ref_cases.py -> tests/golden/frame_ref.npz.  Run where the reference checkout is;
the replaying tests (tests/test_frame_ref_golden.py) need neither the checkout nor the libraries."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from tests import frame_ref_cases as C  # noqa: E402


def main(path=None):
    if O.ref_frame_lib() is None or O.ref_mappoint_lib() is None:
        sys.exit("oracle/_ref/libref_frame.so / libref_mappoint.so missing: build them from the reference checkout first (make -C oracle)")
    k, d, sf = C.frame(O.Extractor(1000, 1.2, 8, 20, 7))
    out = {}
    idx, cnt = [], []
    with O.reference_frame():
        for (x, y, r, lo, hi) in C.windows():
            g = O.features_in_area(k, sf, C.W, C.H, x, y, r, lo, hi)
            idx.append(g); cnt.append(len(g))
        fin = C.frustum_inputs(k, sf)
        for limit in (0.5, 0.9):
            res = O.is_in_frustum(k, d, sf, C.W, C.H, C.CAM, *fin, limit)
            for name, a in C.frustum_reduce(res).items():
                out["frustum%g_%s" % (limit, name)] = a
    out["fia_cnt"] = np.array(cnt, np.int32)
    out["fia_idx"] = np.concatenate(idx).astype(np.int32)
    off, desc = C.tracks()
    with O.reference_mappoint():
        best = O.distinctive_descriptors(off, desc)
    out["distinctive_desc"] = np.stack([desc[off[p] + best[p]] for p in range(len(off) - 1)])     # the reference keeps the winning DESCRIPTOR
    path = path or os.path.join(ROOT, "tests", "golden", "frame_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "windows", len(cnt), "indices", int(out["fia_cnt"].sum()), "in view", int(out["frustum0.5_iv"].sum()), int(out["frustum0.9_iv"].sum()))


if __name__ == "__main__":
    main()
