#!/usr/bin/env python3
"""tools/make_golden_align_ref.py -- what THE REFERENCE'S OWN SparseImgAlign (src/SparseImageAlign.cc + include/NLSSolver_impl.hpp in
oracle/_ref/libref_orbmatcher.so) returns on the four scenes of tests/align_ref_cases.py -> tests/golden/align_ref.npz (return value, SE3, number of
linearisations, final chi2, Hessian).  Run where the reference checkout is; the replaying tests need neither the checkout nor the library."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from tests import align_ref_cases as A  # noqa: E402


def main(path=None):
    if O.ref_matcher_lib() is None:
        sys.exit("oracle/_ref/libref_orbmatcher.so is missing: build it from the reference checkout first (make -C oracle ref_matcher)")
    out = {}
    for j in range(len(A.CASES)):
        oex = O.Extractor(600, 1.2, 8, 20, 7)
        imA, imB, k, world, valid, outl, max_level, min_level = A.scene(j, oex)
        with O.reference_matcher():
            ret, T, info, Hm = O.sparse_img_align(k, world, A.IDENT, oex.pyramid(imA), A.IDENT, oex.pyramid(imB), oex.tables()["inv_scale"], A.CAM, max_level, min_level, 10,
                                                  mp_valid=valid, outlier=outl)
        out["ret%d" % j], out["T%d" % j], out["info%d" % j], out["H%d" % j] = np.int64(ret), np.asarray(T, np.float32), np.asarray(info, np.float32), np.asarray(Hm, np.float32)
        print("scene", j, "ret", ret, "T", np.asarray(T), "info", np.asarray(info))
    path = path or os.path.join(ROOT, "tests", "golden", "align_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
