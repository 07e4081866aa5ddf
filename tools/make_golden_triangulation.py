#!/usr/bin/env python3
"""tools/make_golden_triangulation.py -- golden vectors of ORBmatcher::SearchForTriangulation from THE REFERENCE'S OWN CODE: src/ORBmatcher.cc
compiled where it lies into oracle/_ref/libref_orbmatcher.so (oracle/Makefile, target ref_matcher) is run on the cases of tests/tri_cases.py
and its surviving pairs are written to tests/golden/triangulation_ref.npz -- what neither the oracle nor the device can be wrong about
together.  Needs the reference checkout (run in the build container); the tests that read the file need neither.
Inputs are not stored: they are the oracle extractor's keypoints / descriptors of two synthetic frames (bit-exact on the device, test_gpu_golden /
test_gpu_extract); a digest of them is, so that a drift of the inputs fails as such."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from orb_ygz_slam_amd.scene import synth_frame  # noqa: E402
from tests.tri_cases import cases  # noqa: E402

W, H = 752, 480


def frames():
    base = synth_frame(50, W + 16, H + 16)
    return base[8:8 + H, 8:8 + W], base[10:10 + H, 5:5 + W]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main(path=None):
    if O.ref_matcher_lib() is None:
        sys.exit("oracle/_ref/libref_orbmatcher.so is missing: build it from the reference checkout first (make -C oracle ref_matcher)")
    a, b = frames()
    oex = O.Extractor(1000, 1.2, 8, 20, 7)
    ka, da = oex.extract(a)
    kb, db = oex.extract(b)
    sf = oex.tables()["scale"]
    out = {"inputs_sha256": np.array(digest(ka, da, kb, db)), "labels": []}
    for label, kw in cases(ka, da, kb, db):
        with O.reference_matcher():
            n, m = O.search_for_triangulation(scale_factors2=sf, level_sigma2_2=(sf * sf).astype(np.float32), **kw)
        out["labels"].append(label)
        out["n_" + label] = np.int32(n)
        out["m_" + label] = m.astype(np.int32)
    out["labels"] = np.array(out["labels"])
    path = path or os.path.join(ROOT, "tests", "golden", "triangulation_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {l: int(out["n_" + l]) for l in out["labels"]})


if __name__ == "__main__":
    main()
