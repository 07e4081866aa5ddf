#!/usr/bin/env python3
"""tools/make_golden_direct_ref.py -- what THE REFERENCE'S OWN ORBmatcher::FindDirectProjection (+ src/Align.cc, oracle/_ref/libref_orbmatcher.so) returns
for every keypoint of the two scenes of tests/direct_ref_cases.py -> tests/golden/direct_ref.npz (refined pixel bit patterns, search level, success
flag, warped 10 x 10 patch).  Run where the reference checkout is; the replaying tests need neither the checkout nor the library."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from tests import direct_ref_cases as D  # noqa: E402
from tests.test_ref_matcher import _ref_find_direct_projection_batch  # noqa: E402


def main(path=None):
    if O.ref_matcher_lib() is None:
        sys.exit("oracle/_ref/libref_orbmatcher.so is missing: build it from the reference checkout first (make -C oracle ref_matcher)")
    out = {}
    for j in range(len(D.SCENES)):
        oex = O.Extractor(1000, 1.2, 8, 20, 7)
        A, B, cur7, slot, ref7, ka, world, px0 = D.scene(j, oex)
        px, sl, ok, pt = _ref_find_direct_projection_batch(oex, [A], B, cur7, D.CAM, slot, ref7, ka, world, px0)
        out["px%d" % j], out["level%d" % j], out["ok%d" % j], out["patch%d" % j] = px.astype(np.float32), sl.astype(np.int32), ok.astype(np.uint8), pt.astype(np.uint8)
        print("scene", j, len(ka), "candidates,", int(ok.sum()), "aligned")
    path = path or os.path.join(ROOT, "tests", "golden", "direct_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
