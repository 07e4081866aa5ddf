"""Scenarios shared by tools/make_golden_stereo_bow_ref.py -- which runs them through THE REFERENCE'S OWN Frame::ComputeStereoMatches (src/Frame.cc in
oracle/_ref/libref_frame.so) and DBoW2 (Thirdparty/DBoW2 in oracle/_ref/libref_dbow2.so: text loader + ORBVocabulary::transform) and commits what they
returned as tests/golden/stereo_bow_ref.npz -- and by the tests that hold the oracle (CPU tier) and the device (GPU tier) to those bytes."""
import numpy as np

STEREO = ((3, (752, 480), 1200, 8, 1.2, (0.11, 47.9)), (5, (640, 480), 800, 6, 1.2, (0.12, 40.0)), (8, (752, 480), 2000, 8, 1.2, (0.11, 47.9)))
BOW = ((10, 3, 1, 1), (10, 3, 2, 2), (10, 4, 2, 3), (4, 5, 4, 4), (10, 3, 4, 5), (7, 4, 3, 6))      # (k, L, levelsup, seed)


def bow_descs(voc, n, seed):
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc["is_leaf"])[0]
    d = voc["desc"][rng.choice(leaves, n)].copy()               # near some word, a few bits off
    for i in range(n):
        for b in rng.integers(0, 256, rng.integers(0, 40)):
            d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    d[: n // 10] = rng.integers(0, 256, (n // 10, 32), dtype=np.uint8)
    return d


def fv_flat(fv):
    """FeatureVector (node id -> feature indices) as one int array: id, count, indices..., in ascending id order."""
    out = []
    for key in sorted(fv):
        out.append(int(key)); out.append(len(fv[key])); out.extend(int(x) for x in fv[key])
    return np.array(out, np.int64)
