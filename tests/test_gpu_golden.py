"""GPU tier against COMMITTED bytes: the HIP path is driven through the same scenario script as tools/make_golden_paths.py (an adapter
gives the C-ABI binding the oracle module's call signatures) and must reproduce tests/golden/paths_golden.npz -- digests for every
integer / bit-exact result, 1e-5 on the aligner's SE3."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


class _GpuAsOracle:
    """The oracle module's interface (the part make_golden_paths.cases uses) on top of orb_ygz_slam_amd.capi."""

    def __init__(self):
        from orb_ygz_slam_amd import Extractor as GpuExtractor, make_camera
        self._mk, self._cam = GpuExtractor, make_camera
        self.ex = None

    def Extractor(self, nf, sf, nl, ini, mn):
        outer = self
        g = self._mk(nf, sf, nl, ini, mn, max_width=752, max_height=480, max_batch=2)
        self.ex = g

        class E:
            def extract(self, img): return g.extract(img)
            def tables(self): return g.tables()
            def level_size(self, w, h, l): return g.level_size(w, h, l)
            def extract_dso(self, img, existing=None, grid_size=-1): return g.extract_dso(img, existing=existing, grid_size=grid_size)
            def compute_stereo_matches(self, *a): return g.compute_stereo_matches(*a)
            def pyramid(self, img): return g.compute_pyramid(img)

            def find_direct_projection_batch(self, refs, cur, T7, cam, slot, refT, kp, world, px):
                g.image_cache_reserve(len(refs) + 1, cur.shape[1], cur.shape[0])
                for i, r in enumerate(refs):
                    g.image_cache_put(i, r)
                g.image_cache_put(len(refs), cur)
                return g.find_direct_projection_batch(outer._cam(cur.shape[1], cur.shape[0]), len(refs), T7, slot, refT, kp, world, px, want_patches=True)
        return E()

    def search_by_projection_last(self, kb, db, sf, w, h, cam, ka, world, da, Rcw, tcw, Rlw, tlw, th):
        return self.ex.search_by_projection_last(self._cam(w, h), kb, db, ka, world, da, Rcw, tcw, Rlw, tlw, th, scale_factors=sf)

    def search_by_projection_mappoints(self, kb, db, sf, w, h, cam, tiv, px, py, vc, lvl, da, th, chk, ratio):
        return self.ex.search_by_projection_mappoints(self._cam(w, h), kb, db, tiv, px, py, vc, lvl, da, th, chk, ratio, scale_factors=sf)

    def search_for_initialization(self, ka, da, kb, db, sf, w, h, cam, prev, win, ratio, ori):
        return self.ex.search_for_initialization(self._cam(w, h), ka, da, kb, db, prev, win, ratio, ori, scale_factors=sf)

    def search_by_bow(self, *a):
        return self.ex.search_by_bow(*a)

    def is_in_frustum(self, kb, db, sf, w, h, cam, world, normal, mx, mn, mf, Rcw, tcw, Ow, lsf, lim):
        return self.ex.is_in_frustum_batch(self._cam(w, h), world, normal, mx, mn, mf, Rcw, tcw, Ow, lsf, lim)

    def distinctive_descriptors(self, off, desc):
        return self.ex.distinctive_descriptors_batch(off, desc)

    def sparse_img_align(self, k, wp, Tr, pa, Tc, pb, inv, cam, max_level, min_level):
        return self.ex.sia_run(self._cam(752, 480), k, wp, Tr, pa, Tc, pb, inv, max_level, min_level)


def test_gpu_reproduces_committed_goldens():
    from make_golden_paths import cases
    from tests.test_oracle_paths_golden import check_against_golden
    check_against_golden(cases(_GpuAsOracle()), float_tol={("align", "T"): 1e-5})


def test_gpu_reproduces_extract_goldens():
    """HIP extractor against the committed extract_golden.npz (the FAST-10 fixture with the reference library's own 167-corner known
    answer is replayed by tests/test_gpu_fast10.py)."""
    import hashlib
    from orb_ygz_slam_amd import Extractor
    from orb_ygz_slam_amd.synth import synth_frame
    from tests.test_oracle_golden import CASES, GOLD as EG
    for name, w, h, seed, cfg in CASES:
        img = synth_frame(seed, w, h)
        for mode in (0, 1, 2):   # ygzf_cv_mode: the three GaussianBlur generations
            k, d = Extractor(*cfg, max_width=w, max_height=h, max_batch=1, cv_mode=mode).extract(img)
            assert len(k) == int(EG[name + "_n"][0])
            assert hashlib.sha256(k.tobytes() + d.tobytes()).digest() == EG["%s_m%d_sha" % (name, mode)].tobytes(), (name, mode)
