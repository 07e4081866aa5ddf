"""k_fast_tab -- the FAST cell loop started from host-built per-cell records with the window staged by LDS-DMA (16-byte pieces from
byte-unaligned addresses) -- must return exactly what the register-staging kernel k_fast_quads and the oracle return: candidates (x, y,
score, order) of every level, keypoints, descriptors; on both threshold plans, on every window alignment, for cells clipped at the right /
bottom border, for cells too wide for its window pitch (fallback), launch after launch and across geometry changes."""
import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame
from tests.test_gpu_fast_plans import _images

pytestmark = pytest.mark.gpu


def _cmp(ex, oex, img, frame=0, what=""):
    ok, od = oex.extract(img)
    for l in range(oex.nlevels):
        xs, ys, sc = oex.cell_candidates(l)
        gx, gy, gs = ex.batch_fetch_candidates(frame, l)
        assert len(gx) == len(xs), (what, l, len(gx), len(xs))
        assert (gx == xs).all() and (gy == ys).all() and (gs == sc).all(), (what, l)
    k, d = ex.batch_fetch(frame)
    assert len(k) == len(ok) and (k == ok).all() and (d == od).all(), what


@pytest.mark.parametrize("plan", [1, 2])
@pytest.mark.parametrize("ini,mn", [(20, 7), (12, 12), (40, 5)])
def test_table_kernel_equals_the_oracle_on_every_content(oracle, plan, ini, mn):
    from orb_ygz_slam_amd import Extractor
    w, h = 640, 480
    ex = Extractor(1000, 1.2, 8, ini, mn, max_width=w, max_height=h, max_batch=1)
    ex.set_fast_plan(plan)
    ex.set_fast_kernel(2)
    oex = oracle.Extractor(1000, 1.2, 8, ini, mn)
    for name, img in _images(w, h).items():
        ex.extract_batch_host(img[None])
        _cmp(ex, oex, img, 0, (name, plan))


# widths chosen so that the cell width and the byte alignment (iniX - 1) & 15 of the windows vary; 1.5 / 2.0 pyramids reach small levels
# with single cells; the 1241x376 KITTI shape has the odd width the API re-pitches; 91x91 and 123x200 have cells of 59 / 45 px, wider than
# the table kernel's window (the library must fall back to k_fast_quads on its own)
@pytest.mark.parametrize("w,h,sf,nl", [(752, 480, 1.2, 8), (640, 480, 1.2, 8), (631, 397, 1.2, 8), (517, 333, 1.3, 6), (1241, 376, 1.2, 8),
                                       (401, 303, 1.5, 5), (322, 243, 2.0, 3), (96, 128, 1.2, 3), (91, 91, 1.2, 2), (123, 200, 1.2, 3), (1920, 1080, 1.2, 8)])
def test_table_kernel_sizes(oracle, w, h, sf, nl):
    from orb_ygz_slam_amd import Extractor
    ex = Extractor(1500, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=2)
    ex.set_fast_kernel(2)
    oex = oracle.Extractor(1500, sf, nl, 20, 7)
    imgs = np.stack([synth_frame(100 + w, w, h), synth_frame(200 + h, w, h)])
    for plan in (1, 2):
        ex.set_fast_plan(plan)
        ex.extract_batch_host(imgs)
        for f in range(2):
            _cmp(ex, oex, imgs[f], f, (w, h, sf, nl, plan, f))


def test_table_kernel_launch_after_launch(oracle):
    """batches of different sizes on one context, then the other kernel on the same context"""
    from orb_ygz_slam_amd import Extractor
    w, h = 752, 480
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=24)
    ex.set_fast_kernel(2)
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    frames = np.stack([synth_frame(300 + i, w, h) for i in range(24)])
    want = [oex.extract(frames[i]) for i in range(24)]
    for n in (24, 1, 7, 24, 2, 1, 16):
        ex.extract_batch_host(frames[:n])
        for f in range(n):
            k, d = ex.batch_fetch(f)
            assert len(k) == len(want[f][0]) and (k == want[f][0]).all() and (d == want[f][1]).all(), (n, f)
    ex.set_fast_kernel(1)                                      # and the register-staging kernel on the same context still agrees
    ex.extract_batch_host(frames[:5])
    for f in range(5):
        k, d = ex.batch_fetch(f)
        assert (k == want[f][0]).all() and (d == want[f][1]).all()


def test_automatic_choice_is_result_neutral(oracle):
    from orb_ygz_slam_amd import Extractor
    w, h = 640, 480
    frames = np.stack([synth_frame(500 + i, w, h) for i in range(32)])
    res = []
    for kern in (0, 1, 2):
        ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=32)
        ex.set_fast_kernel(kern)
        ex.extract_batch_host(frames)
        res.append([ex.batch_fetch(f) for f in range(32)])
    for f in range(32):
        for r in res[1:]:
            assert (r[f][0] == res[0][f][0]).all() and (r[f][1] == res[0][f][1]).all()
