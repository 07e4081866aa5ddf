#!/usr/bin/env python3
"""tools/make_golden_matcher_ref.py -- golden vectors of the four Tracking-side searches from THE REFERENCE'S OWN src/ORBmatcher.cc
(oracle/_ref/libref_orbmatcher.so, built by oracle/Makefile from the checkout) on the scenarios of tests/matcher_ref_cases.py
-> tests/golden/matcher_ref.npz.  Run where the reference checkout is; the tests that read the file (oracle: CPU tier, device: GPU tier) need
neither the checkout nor the library."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from tests import matcher_ref_cases as C  # noqa: E402


def main(path=None):
    if O.ref_matcher_lib() is None:
        sys.exit("oracle/_ref/libref_orbmatcher.so is missing: build it from the reference checkout first (make -C oracle ref_matcher)")
    ka, da, kb, db, sf = C.inputs(O.Extractor(1000, 1.2, 8, 20, 7))
    h = hashlib.sha256()
    for a in (ka, da, kb, db):
        h.update(np.ascontiguousarray(a).tobytes())
    out = {"inputs_sha256": np.array(h.hexdigest())}
    names = []
    for name, fn, args, kw in C.cases(ka, da, kb, db, sf):
        with O.reference_matcher():
            res = getattr(O, fn)(*args, **kw)
        for k, v in C.reduce(fn, res).items():
            out[name + "_" + k] = v
        names.append(name)
        print(name, int(res[0]))
    out["names"] = np.array(names)
    path = path or os.path.join(ROOT, "tests", "golden", "matcher_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
