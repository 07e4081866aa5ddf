"""tests/golden/stereo_bow_ref.npz holds what THE REFERENCE'S OWN Frame::ComputeStereoMatches (src/Frame.cc) and DBoW2 (text loader +
ORBVocabulary::transform, the call behind Frame::ComputeBoW) returned on the scenarios of tests/stereo_bow_ref_cases.py
(tools/make_golden_stereo_bow_ref.py, run where the checkout is): mvuRight / mvDepth bit patterns of three stereo pairs, BowVector ids / tf-idf
doubles / FeatureVector of six vocabularies.  The oracle (CPU tier) and the device (GPU tier) must reproduce those bytes wherever they run."""
import os

import numpy as np
import pytest

from orb_ygz_slam_amd.scene import stereo_scene
from tests import stereo_bow_ref_cases as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stereo_bow_ref.npz")


def _check(oracle, make_extractor, descend):
    g = np.load(GOLD)
    for j, (seed, (w, h), nf, nl, sf, (mb, mbf)) in enumerate(S.STEREO):
        left, right, _, _ = stereo_scene(seed, w, h)
        ex = make_extractor(nf, sf, nl, w, h)
        kl, dl = ex.extract(left)
        kr, dr = ex.extract(right)
        ur, dp = ex.compute_stereo_matches(left, right, kl, dl, kr, dr, mb, mbf)
        assert np.array_equal(np.asarray(ur, np.float32).view(np.uint32), g["stereo%d_ur" % j].view(np.uint32)), j
        assert np.array_equal(np.asarray(dp, np.float32).view(np.uint32), g["stereo%d_depth" % j].view(np.uint32)), j
    for j, (k, L, levelsup, seed) in enumerate(S.BOW):
        voc = oracle.make_vocabulary(seed, k, L)
        d = S.bow_descs(voc, 700, seed + 100)
        leaf, nid = descend(voc, L, d, levelsup)
        ids, vals, fv = oracle.bow_vectors(voc, leaf, nid)      # the maps a host assembles from the descent (host/ORBVocabularyDevice.cc does the same)
        assert np.array_equal(np.asarray(ids), g["bow%d_ids" % j]), j
        assert np.array_equal(np.asarray(vals, np.float64).view(np.uint64), g["bow%d_vals" % j].view(np.uint64)), j
        assert np.array_equal(S.fv_flat(fv), g["bow%d_fv" % j]), j


def test_oracle_reproduces_the_reference_stereo_and_bow_golden(oracle):
    _check(oracle, lambda nf, sf, nl, w, h: oracle.Extractor(nf, sf, nl, 20, 7), lambda voc, L, d, levelsup: oracle.bow_descend(voc, d, levelsup))


@pytest.mark.gpu
def test_device_reproduces_the_reference_stereo_and_bow_golden(oracle):
    from orb_ygz_slam_amd import Extractor
    ex0 = Extractor(500, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)

    def descend(voc, L, d, levelsup):
        ex0.vocabulary_set(voc["parent"], voc["desc"], L)
        return ex0.bow_transform(d, levelsup)

    _check(oracle, lambda nf, sf, nl, w, h: Extractor(nf, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=2), descend)
