#!/usr/bin/env python3
"""tools/make_golden_stereo_bow_ref.py -- what THE REFERENCE'S OWN Frame::ComputeStereoMatches (oracle/_ref/libref_frame.so) and DBoW2 transform
(oracle/_ref/libref_dbow2.so) return on the scenarios of tests/stereo_bow_ref_cases.py -> tests/golden/stereo_bow_ref.npz.  Run where the reference
checkout is; the replaying tests (tests/test_stereo_bow_ref_golden.py) need neither the checkout nor the libraries."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from orb_ygz_slam_amd.scene import stereo_scene  # noqa: E402
from tests import stereo_bow_ref_cases as S  # noqa: E402


def main(path=None):
    L = O.ref_frame_lib()
    if L is None or O.ref_dbow2_lib() is None:
        sys.exit("oracle/_ref/libref_frame.so / libref_dbow2.so missing: build them from the reference checkout first (make -C oracle)")
    L.yr_stereo_config.argtypes = [C.c_int, C.c_float]
    L.yo_compute_stereo_matches.restype = None
    L.yo_compute_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    out = {}
    for j, (seed, (w, h), nf, nl, sf, (mb, mbf)) in enumerate(S.STEREO):
        left, right, _, _ = stereo_scene(seed, w, h)
        ex = O.Extractor(nf, sf, nl, 20, 7)
        kl, dl = ex.extract(left)
        kr, dr = ex.extract(right)
        L.yr_stereo_config(nl, sf)
        il, ir = np.ascontiguousarray(left), np.ascontiguousarray(right)
        ur, dp = np.zeros(len(kl), np.float32), np.zeros(len(kl), np.float32)
        L.yo_compute_stereo_matches(None, p(il), p(ir), w, h, len(kl), p(kl), p(dl), len(kr), p(kr), p(dr), mb, mbf, p(ur), p(dp))
        out["stereo%d_ur" % j], out["stereo%d_depth" % j] = ur, dp
        print("stereo", j, int((ur >= 0).sum()), "matches of", len(kl))
    with tempfile.TemporaryDirectory() as tmp:
        for j, (k, Lv, levelsup, seed) in enumerate(S.BOW):
            voc = O.make_vocabulary(seed, k, Lv)
            vpath = os.path.join(tmp, "voc%d.txt" % j)
            O.write_vocabulary_text(voc, vpath)
            ids, vals, fv = O.RefVocabulary(vpath).transform(S.bow_descs(voc, 700, seed + 100), levelsup)
            out["bow%d_ids" % j], out["bow%d_vals" % j], out["bow%d_fv" % j] = np.asarray(ids), np.asarray(vals, np.float64), S.fv_flat(fv)
            print("bow", j, len(ids), "words")
    path = path or os.path.join(ROOT, "tests", "golden", "stereo_bow_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
