"""The pose algebra under the aligner (SE3f::exp, operator*, inverse, the action on a point) pinned to the REFERENCE'S OWN Sophus: the oracle's
restatement (oracle/oracle_align.cpp -- what the aligner pin of tests/test_ref_matcher.py runs through, and what the HIP kernel's se3_device.h is
compared with) against Thirdparty/sophus/sophus/se3.hpp + so3.hpp compiled where they lie (oracle/ref_sophus_capi.cpp, `make -C oracle ref_sophus`)
over oracle/ref_shim/eigen_min, bit for bit on random and edge-case inputs.  Eigen is not in the reference checkout: its quaternion / small-matrix
primitives remain the stand-in of that one header; every Sophus line executed here is the reference's own (se3.hpp:159-171, 267-271, 406-428,
so3.hpp:268-276, 425-456)."""
import ctypes as C

import numpy as np
import pytest

from tests.conftest import ROOT  # noqa: F401


def _ref(oracle):
    L = oracle.ref_sophus_lib()
    if L is None:
        pytest.skip("oracle/_ref/libref_sophus.so not built (no reference checkout)")
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _poses(rng, n):
    out = []
    for _ in range(n):
        a = np.concatenate([rng.uniform(-2, 2, 3), rng.uniform(-1, 1, 3) * rng.choice([1e-7, 1e-3, 0.3, 3.0])]).astype(np.float32)
        out.append(a)
    return out


def test_exp_is_sophus_exp(oracle):
    L = _ref(oracle)
    rng = np.random.default_rng(5)
    cases = _poses(rng, 4000)
    cases += [np.zeros(6, np.float32), np.array([1, 2, 3, 0, 0, 0], np.float32), np.array([0, 0, 0, 1e-6, 0, 0], np.float32),       # below Sophus' epsilon
              np.array([0.1, -0.2, 0.3, 9.9e-6, 0, 0], np.float32), np.array([0.1, -0.2, 0.3, 1.01e-5, 0, 0], np.float32),       # either side of it
              np.array([0, 0, 0, np.pi, 0, 0], np.float32), np.array([1, 1, 1, 2.2, -2.2, 2.2], np.float32)]
    for a in cases:
        want = np.zeros(7, np.float32)
        L.ref_sophus_exp(_p(a), _p(want))
        got = oracle.se3_exp(a)
        assert np.array_equal(_bits(got), _bits(want)), (a, got, want)


def test_mul_inverse_act_are_sophus(oracle):
    L = _ref(oracle)
    rng = np.random.default_rng(6)
    poses = [oracle.se3_exp(a) for a in _poses(rng, 600)]
    # poses that are NOT unit (what accumulated updates look like before operator*= renormalises) take the same path in both
    poses += [np.concatenate([p[:4] * np.float32(1.0003), p[4:]]).astype(np.float32) for p in poses[:50]]
    for i in range(0, len(poses) - 1):
        a, b = poses[i], poses[(7 * i + 3) % len(poses)]
        want = np.zeros(7, np.float32)
        L.ref_sophus_mul(_p(a), _p(b), _p(want))
        assert np.array_equal(_bits(oracle.se3_mul(a, b)), _bits(want)), (a, b)
        L.ref_sophus_inverse(_p(a), _p(want))
        assert np.array_equal(_bits(oracle.se3_inverse(a)), _bits(want)), a
        pt = rng.uniform(-5, 5, 3).astype(np.float32)
        w3 = np.zeros(3, np.float32)
        L.ref_sophus_act(_p(a), _p(pt), _p(w3))
        assert np.array_equal(_bits(oracle.se3_act(a, pt)), _bits(w3)), (a, pt)


def test_gauss_newton_update_chain_is_sophus(oracle):
    """the aligner's update T <- T * exp(-x) (src/SparseImageAlign.cc:240-244) iterated: no drift between the two over forty steps"""
    L = _ref(oracle)
    rng = np.random.default_rng(7)
    To = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    Tr = To.copy()
    for _ in range(40):
        x = (rng.uniform(-1, 1, 6) * 0.01).astype(np.float32)
        eo = oracle.se3_exp(-x)
        er = np.zeros(7, np.float32)
        L.ref_sophus_exp(_p(-x), _p(er))
        assert np.array_equal(_bits(eo), _bits(er))
        To = oracle.se3_mul(To, eo)
        nr = np.zeros(7, np.float32)
        L.ref_sophus_mul(_p(Tr), _p(er), _p(nr))
        Tr = nr
        assert np.array_equal(_bits(To), _bits(Tr))
