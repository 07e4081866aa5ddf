"""tools/ref_fuzz.py -- one-off wide sweeps of the oracle against the reference's own source (oracle/_ref/*.so): more seeds than the committed
tests/test_ref_*.py run by default.  CPU only; needs the reference checkout (or prebuilt oracle/_ref).  Round 1: 108 extractor configurations,
52 matcher scenarios, 20 stereo pairs, 20 aligner runs -- zero differences."""
import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import test_ref_extractor as E, test_ref_matcher as M
bad=0
for seed in range(12, 120):
    try:
        E.test_fuzz_sizes_and_configs_equal_reference(seed)
    except AssertionError as e:
        bad+=1; print("extractor seed",seed,"FAIL",str(e)[:200])
print("extractor fuzz done, failures:",bad)
bad=0
for seed in range(8, 60):
    try:
        M.test_fuzz_projection_searches_equal_reference(seed)
    except AssertionError as e:
        bad+=1; print("matcher seed",seed,"FAIL",str(e)[:200])
    except BaseException as e:
        if 'Skipped' in type(e).__name__: continue
        raise
print("matcher fuzz done, failures:",bad)

# ---- Frame::ComputeStereoMatches and SparseImgAlign
import ctypes as C
from oracle import oracle_py as O
from orb_ygz_slam_amd.scene import stereo_scene, synth_frame, two_view_scene
L = O.ref_frame_lib()
L.yr_stereo_config.argtypes = [C.c_int, C.c_float]
L.yo_compute_stereo_matches.restype = None
L.yo_compute_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
p = lambda a: a.ctypes.data_as(C.c_void_p)
bad = 0
rng = np.random.default_rng(0)
for seed in range(20, 40):
    w, h = int(rng.choice([640, 752, 512])), int(rng.choice([480, 400]))
    nf, nl = int(rng.choice([500, 1000, 1800])), int(rng.choice([5, 8]))
    left, right, _, _ = stereo_scene(seed, w, h)
    ex = O.Extractor(nf, 1.2, nl, 20, 7)
    kl, dl = ex.extract(left); kr, dr = ex.extract(right)
    e_ur, e_dp = ex.compute_stereo_matches(left, right, kl, dl, kr, dr, 0.11, 47.9)
    L.yr_stereo_config(nl, 1.2)
    ur, dp = np.zeros(len(kl), np.float32), np.zeros(len(kl), np.float32)
    L.yo_compute_stereo_matches(None, p(np.ascontiguousarray(left)), p(np.ascontiguousarray(right)), w, h, len(kl), p(kl), p(dl), len(kr), p(kr), p(dr), 0.11, 47.9, p(ur), p(dp))
    if not (np.array_equal(ur.view(np.uint32), e_ur.view(np.uint32)) and np.array_equal(dp.view(np.uint32), e_dp.view(np.uint32))):
        bad += 1; print("stereo seed", seed, "DIFF", (ur != e_ur).sum())
print("stereo fuzz failures", bad)
# SIA fuzz
import test_ref_matcher as M
bad = 0
ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
for seed in range(40, 60):
    rv = tuple(rng.uniform(-0.01, 0.01, 3)); tr = tuple(rng.uniform(-0.05, 0.05, 3))
    A, B, _, bp = two_view_scene(seed, 752, 480, M.CAM, Z=float(rng.uniform(2, 6)), rotvec=rv, trans=tr)
    oex = O.Extractor(int(rng.choice([300, 600, 1000])), 1.2, 8, 20, 7)
    k, _ = oex.extract(A); world = bp(k["x"], k["y"])
    pa, pb = oex.pyramid(A), oex.pyramid(B); inv = oex.tables()["inv_scale"]
    mx, mn = int(rng.integers(2, 6)), int(rng.integers(0, 2))
    args = (k, world, ident, pa, ident, pb, inv, M.CAM, mx, mn, 10)
    e = O.sparse_img_align(*args)
    with O.reference_matcher():
        r = O.sparse_img_align(*args)
    if not (r[0] == e[0] and (r[1].view(np.uint32) == e[1].view(np.uint32)).all() and r[2][0] == e[2][0]):
        bad += 1; print("sia seed", seed, "DIFF", r[0], e[0], r[1], e[1])
print("sia fuzz failures", bad)
