"""The oracle against THE REFERENCE'S OWN src/MapPoint.cc (real include/MapPoint.h; KeyFrame / Frame / Map are plain-data stubs of
oracle/ref_shim/), compiled where it lies into oracle/_ref/libref_mappoint.so: MapPoint::ComputeDistinctiveDescriptors,
MapPoint::PredictScale (used by SearchByProjection(Cur, KF) and Frame::isInFrustum) and the distance-invariance range that
UpdateNormalAndDepth sets.  CPU tier; skipped where the library was never built."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_py as O

pytestmark = pytest.mark.skipif(O.ref_mappoint_lib() is None, reason="oracle/_ref/libref_mappoint.so not built (reference checkout absent)")


def test_distinctive_descriptors_equal_reference():
    rng = np.random.default_rng(17)
    sizes = [1, 2, 3, 4, 5, 8, 13, 21, 40, 2, 7, 64]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    total = int(off[-1])
    # observations of one point are noisy copies of a common descriptor (ties in the median are frequent, as with real tracks)
    desc = np.zeros((total, 32), np.uint8)
    for p, n in enumerate(sizes):
        base = rng.integers(0, 256, 32).astype(np.uint8)
        for i in range(n):
            flips = np.packbits(rng.uniform(size=256) < rng.choice([0.02, 0.1, 0.3]))
            desc[off[p] + i] = base ^ flips
    e = O.distinctive_descriptors(off, desc)
    with O.reference_mappoint():
        r = O.distinctive_descriptors(off, desc)
    # the reference keeps the winning DESCRIPTOR; identical observations make the index ambiguous -> compare the descriptors
    for p in range(len(sizes)):
        assert (desc[off[p] + r[p]] == desc[off[p] + e[p]]).all(), (p, r[p], e[p])
    assert (r == e).all()


def test_predict_scale_equals_reference():
    rng = np.random.default_rng(3)
    ratio = np.concatenate([np.float32(1.2) ** np.arange(-3, 12, dtype=np.float32), rng.uniform(0.05, 8.0, 4000).astype(np.float32),
                            np.nextafter(np.float32(1.2) ** np.arange(0, 8, dtype=np.float32), np.float32(10))]).astype(np.float32)
    for sf, nl in ((1.2, 8), (1.5, 5), (2.0, 4), (1.1, 10)):
        logsf = np.log(np.float32(sf))
        e = O.predict_scale(ratio, logsf, nl)
        with O.reference_mappoint():
            r = O.predict_scale(ratio, logsf, nl)
        assert (r == e).all(), (sf, nl)


def test_distance_invariance_range():
    """UpdateNormalAndDepth -> GetMin/MaxDistanceInvariance: what the matcher tests build by hand (0.8 * max / scale[n-1], 1.2 * |PC| * scale[level])."""
    L = O.ref_mappoint_lib()
    rng = np.random.default_rng(5)
    n = 500
    pos = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    ow = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    lvl = rng.integers(0, 8, n).astype(np.int32)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))])).astype(np.float32)      # mvScaleFactor[i] = mvScaleFactor[i-1] * 1.2f
    omin, omax = np.zeros(n, np.float32), np.zeros(n, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.yr_distance_invariance.restype = None
    L.yr_distance_invariance.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.yr_distance_invariance(n, p(pos), p(ow), p(lvl), p(sf), 8, p(omin), p(omax))
    pc = pos - ow
    dist = np.sqrt((pc[:, 0] * pc[:, 0] + pc[:, 1] * pc[:, 1]) + pc[:, 2] * pc[:, 2]).astype(np.float32)
    mf_max = (dist * sf[lvl]).astype(np.float32)
    mf_min = (mf_max / sf[7]).astype(np.float32)
    assert (omax == np.float32(1.2) * mf_max).all() and (omin == np.float32(0.8) * mf_min).all()
