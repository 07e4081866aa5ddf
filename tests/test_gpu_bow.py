"""GPU tier: Frame::ComputeBoW's tree descent on the device (ygzf_vocabulary_set / ygzf_bow_transform, k_bow_descend) against the oracle's
restatement of DBoW2::TemplatedVocabulary::transform (pinned to the reference's own DBoW2 on the CPU tier, tests/test_ref_dbow2.py) and,
where oracle/_ref/libref_dbow2.so travelled, against that library directly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _descs(voc, n, seed):
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc["is_leaf"])[0]
    d = voc["desc"][rng.choice(leaves, n)].copy()
    for i in range(n):
        for b in rng.integers(0, 256, rng.integers(0, 40)):
            d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    d[: n // 10] = rng.integers(0, 256, (n // 10, 32), dtype=np.uint8)
    return d


@pytest.mark.parametrize("k,L,levelsup,seed", [(10, 3, 1, 1), (10, 4, 2, 3), (4, 5, 4, 4), (10, 5, 4, 5), (70, 2, 1, 6), (3, 7, 4, 7)])
def test_bow_descent_bit_exact(oracle, tmp_path, k, L, levelsup, seed):
    from orb_ygz_slam_amd import Extractor
    voc = oracle.make_vocabulary(seed, k, L)
    ex = Extractor(max_width=64, max_height=64)
    ex.vocabulary_set(voc["parent"], voc["desc"], L)
    d = _descs(voc, 1500, seed + 50)
    d[5] = voc["desc"][np.nonzero(voc["is_leaf"])[0][3]]                    # an exact word
    leaf, nid = ex.bow_transform(d, levelsup)
    o_leaf, o_nid = oracle.bow_descend(voc, d, levelsup)
    assert (leaf == o_leaf).all() and (nid == o_nid).all()
    assert voc["is_leaf"][leaf].all()
    if oracle.ref_dbow2_lib() is not None and k <= 20:                      # the reference's text loader refuses k > 20
        path = str(tmp_path / "voc.txt")
        oracle.write_vocabulary_text(voc, path)
        r_ids, r_vals, r_fv = oracle.RefVocabulary(path).transform(d, levelsup)
        g_ids, g_vals, g_fv = oracle.bow_vectors(voc, leaf, nid)            # the maps assembled from the DEVICE descent
        assert (g_ids == r_ids).all() and (g_vals.view(np.uint64) == r_vals.view(np.uint64)).all()
        assert sorted(g_fv) == sorted(r_fv) and all((g_fv[key] == r_fv[key]).all() for key in g_fv)


def test_bow_descent_at_orbvoc_size(oracle):
    """The shape of the reference's ORBvoc (k = 10, L = 6: 1 111 111 nodes, 10^6 words, levelsup 4 as Frame::ComputeBoW asks -- the blob itself is
    not shipped, /root/reference/.MISSING_LARGE_BLOBS): the whole tree resident in HBM (36 MB of centroids), 4000 descriptors, leaf and level-2 node of
    every one equal to the oracle's descent; the BowVector / FeatureVector assembled from the device's descent equal the oracle's, doubles bit for bit."""
    from orb_ygz_slam_amd import Extractor
    k, L, levelsup = 10, 6, 4
    voc = oracle.make_vocabulary(11, k, L)
    assert len(voc["parent"]) == 1111111 and int(voc["is_leaf"].sum()) == 1000000
    ex = Extractor(max_width=64, max_height=64)
    ex.vocabulary_set(voc["parent"], voc["desc"], L)
    d = _descs(voc, 4000, 77)
    leaf, nid = ex.bow_transform(d, levelsup)
    o_leaf, o_nid = oracle.bow_descend(voc, d, levelsup)
    assert (leaf == o_leaf).all() and (nid == o_nid).all() and voc["is_leaf"][leaf].all()
    assert len(np.unique(leaf)) > 3000 and len(np.unique(nid)) > 90          # spread over the tree: words of many branches, the 100 level-2 nodes
    g = oracle.bow_vectors(voc, leaf, nid)
    o = oracle.bow_vectors(voc, o_leaf, o_nid)
    assert (g[0] == o[0]).all() and (g[1].view(np.uint64) == o[1].view(np.uint64)).all() and sorted(g[2]) == sorted(o[2])


def test_bow_ties_ragged_and_errors(oracle):
    from orb_ygz_slam_amd import Extractor
    from orb_ygz_slam_amd.capi import YgzfError
    ex = Extractor(max_width=64, max_height=64)
    with pytest.raises(YgzfError):
        ex.bow_transform(np.zeros((4, 32), np.uint8))                       # no vocabulary yet
    voc = oracle.make_vocabulary(9, 3, 3)
    voc["desc"][2] = voc["desc"][1]                                         # identical siblings: the smaller node id wins
    voc["desc"][3] = voc["desc"][1]
    ex.vocabulary_set(voc["parent"], voc["desc"], 3)
    q = np.stack([voc["desc"][1], voc["desc"][3] ^ np.uint8(1)])
    leaf, nid = ex.bow_transform(q, 2)
    o_leaf, o_nid = oracle.bow_descend(voc, q, 2)
    assert (leaf == o_leaf).all() and (nid == o_nid).all() and nid[0] == 1
    voc = oracle.make_vocabulary(10, 3, 3)
    # ragged tree: node 2's subtree removed -> node 2 is a leaf at level 1 (levelsup 1 asks for level 2: defined as that leaf)
    keep = np.ones(len(voc["parent"]), bool)
    par = voc["parent"]
    for i in range(1, len(par)):
        a = i
        while a > 0:
            if par[a] == 2:
                keep[i] = False
            a = par[a]
    remap = np.cumsum(keep) - 1
    rag = dict(k=3, L=3, parent=np.where(par[keep] >= 0, remap[np.maximum(par[keep], 0)], -1).astype(np.int32), desc=voc["desc"][keep],
               weight=voc["weight"][keep])
    rag["is_leaf"] = (np.bincount(rag["parent"][1:], minlength=len(rag["parent"])) == 0).astype(np.uint8)
    ex.vocabulary_set(rag["parent"], rag["desc"], 3)
    d = _descs(voc, 300, 77)
    leaf, nid = ex.bow_transform(d, 1)
    o_leaf, o_nid = oracle.bow_descend(rag, d, 1)
    assert (leaf == o_leaf).all() and (nid == o_nid).all()
    assert (leaf == 2).any()
    assert (ex.bow_transform(np.zeros((0, 32), np.uint8))[0].shape == (0,))
