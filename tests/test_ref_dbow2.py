"""CPU tier: the oracle's restatement of DBoW2's vocabulary transform (oracle_py.bow_descend / bow_vectors) against the REFERENCE's own DBoW2
(Thirdparty/DBoW2, compiled where it lies into oracle/_ref/libref_dbow2.so): generated vocabularies are written in the reference's text
format, read back by its own loadFromTextFile, and ORBVocabulary::transform(features, BowVector, FeatureVector, levelsup) -- the call behind
Frame::ComputeBoW (src/Frame.cc:495-500) -- must return the same word ids, the same tf-idf values bit for bit, and the same FeatureVector."""
import numpy as np
import pytest


def _descs(voc, n, seed):
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc["is_leaf"])[0]
    d = voc["desc"][rng.choice(leaves, n)].copy()               # near some word, a few bits off
    for i in range(n):
        for b in rng.integers(0, 256, rng.integers(0, 40)):
            d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    d[: n // 10] = rng.integers(0, 256, (n // 10, 32), dtype=np.uint8)
    return d


@pytest.mark.parametrize("k,L,levelsup,seed", [(10, 3, 1, 1), (10, 3, 2, 2), (10, 4, 2, 3), (4, 5, 4, 4), (10, 3, 4, 5), (7, 4, 3, 6)])
def test_transform_equals_reference_dbow2(oracle, tmp_path, k, L, levelsup, seed):
    if oracle.ref_dbow2_lib() is None:
        pytest.skip("oracle/_ref/libref_dbow2.so not built (no reference checkout)")
    voc = oracle.make_vocabulary(seed, k, L)
    path = str(tmp_path / "voc.txt")
    oracle.write_vocabulary_text(voc, path)
    ref = oracle.RefVocabulary(path)
    assert ref.size() == int(voc["is_leaf"].sum())
    d = _descs(voc, 700, seed + 100)
    r_ids, r_vals, r_fv = ref.transform(d, levelsup)
    leaf, nid = oracle.bow_descend(voc, d, levelsup)
    o_ids, o_vals, o_fv = oracle.bow_vectors(voc, leaf, nid)
    assert (o_ids == r_ids).all()
    assert (o_vals.view(np.uint64) == r_vals.view(np.uint64)).all()          # same doubles: same summation and normalisation order
    assert sorted(o_fv) == sorted(r_fv)
    for key in o_fv:
        assert (o_fv[key] == r_fv[key]).all()
    assert len(o_ids) > 50 and abs(o_vals.sum() - 1.0) < 1e-12


def test_first_minimum_wins(oracle):
    """Two children at the same distance: the one with the smaller node id is taken (`d < best_d`)."""
    voc = oracle.make_vocabulary(9, 3, 2)
    voc["desc"][2] = voc["desc"][1]                               # children 1 and 2 of the root are identical
    q = voc["desc"][1:2].copy()
    leaf, nid = oracle.bow_descend(voc, q, 1)
    assert voc["parent"][leaf[0]] == 1 and nid[0] == 1
