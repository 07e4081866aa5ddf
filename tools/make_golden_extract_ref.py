#!/usr/bin/env python3
"""tools/make_golden_extract_ref.py -- digests of what THE REFERENCE'S OWN src/ORBextractor.cc returns (oracle/_ref/libref_orbextractor.so, built by
oracle/Makefile from the checkout over the OpenCV stand-in of oracle/ref_shim) for the images and configurations of tests/extract_ref_cases.py
-> tests/golden/extract_ref.npz.  Run where the reference checkout is; the replaying tests (tests/test_extract_ref_golden.py: oracle on the CPU
tier, device on the GPU tier) need neither the checkout nor the library."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from tests import extract_ref_cases as C  # noqa: E402


def main(path=None):
    if O.ref_extractor_lib() is None:
        sys.exit("oracle/_ref/libref_orbextractor.so is missing: build it from the reference checkout first (make -C oracle ref)")
    names, counts, digs = [], [], []
    for name, img, (nf, sf, nl, ini, mn) in C.cases():
        k, d = O.ref_extract(img, nf, sf, nl, ini, mn)
        names.append(name); counts.append(len(k)); digs.append(C.digest(k, d))
        print(name, len(k))
    path = path or os.path.join(ROOT, "tests", "golden", "extract_ref.npz")
    np.savez_compressed(path, names=np.array(names), counts=np.array(counts, np.int32), digests=np.array(digs))
    print("wrote", path)


if __name__ == "__main__":
    main()
