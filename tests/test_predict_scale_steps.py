"""CPU tier (no GPU needed: pure host function of the library): the step table the device uses instead of MapPoint::PredictScale's
ceil(logf(ratio) / logScaleFactor) reproduces that expression for every ratio, including the floats right at the steps."""
import ctypes as C

import numpy as np


def test_predict_scale_steps_reproduce_the_expression(oracle):
    from orb_ygz_slam_amd import load_library
    lib = load_library()
    lib.ygzf_predict_scale_steps.argtypes = [C.c_float, C.c_int, C.c_void_p]
    for sf, L in ((1.2, 8), (1.2, 12), (2.0, 4), (1.1, 16)):
        lsf = np.log(np.float32(sf), dtype=np.float32)
        steps = np.zeros(L, np.float32)
        assert lib.ygzf_predict_scale_steps(C.c_float(float(lsf)), L, steps.ctypes.data_as(C.c_void_p)) == 0
        assert (np.diff(steps[1:]) > 0).all()
        rng = np.random.default_rng(0)
        ratios = np.concatenate([np.exp(rng.uniform(-3, 6, 200000)).astype(np.float32),
                                 steps[1:], np.nextafter(steps[1:], np.float32(0)), np.nextafter(steps[1:], np.float32(np.inf))])
        dev = (ratios[:, None] >= steps[None, 1:]).sum(1)          # what k_frustum computes
        assert (dev == oracle.predict_scale(ratios, lsf, L)).all()   # the oracle evaluates the reference expression with this host's logf
