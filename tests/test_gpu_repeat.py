"""Race hunting: every device path repeated many times on the same inputs must return the same bytes every time (a missing barrier
shows up as an occasional difference).  Parity with the oracle is tested elsewhere; here only run-to-run identity."""
import hashlib

import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame

pytestmark = pytest.mark.gpu


def _digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_every_path_is_run_to_run_identical():
    from bench import make_frames
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    from orb_ygz_slam_amd.scene import stereo_scene, two_view_scene, rotvec_to_quat
    w, h = 752, 480
    frames = make_frames(48, w, h)
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=48)
    cam = make_camera(w, h, mbf=47.9, mb=0.11)
    ref = None
    for rep in range(int(__import__('os').environ.get('YGZF_REPEATS', '30'))):     # extract + match + stereo + align over a batch
        ex.extract_batch_host(frames)
        ex.match_batch_prev(cam, 15.0, True, True, True)
        ex.stereo_batch(0.11, 47.9)
        ex.align_batch_prev(cam, 7, 1, 10)
        k, d, n = ex.batch_fetch_all(48)
        parts = [n, ex.match_counts()]
        for f in range(48):
            parts += [k[f][:n[f]], d[f][:n[f]], ex.match_fetch(f)[0][:n[f]]]
        for p in range(24):
            ur, dp = ex.stereo_fetch(p)
            parts += [ur[:n[2 * p]], dp[:n[2 * p]]]
        for f in range(1, 48):
            parts.append(np.asarray(ex.align_fetch(f)[1]))
        dg = _digest(*parts)
        if rep == 0:
            continue            # the first batch has no carried predecessor for frame 0; from the second on the state is periodic
        ref = ref or dg
        assert dg == ref, "batch pipeline differs in repetition %d" % rep
    # single-frame paths
    img = frames[3]
    ka, da = ex.extract(frames[0])
    kb, db = ex.extract(frames[1])
    n = len(ka)
    rng = np.random.default_rng(0)
    world = np.stack([(ka["x"] - np.float32(EUROC["cx"])) / np.float32(EUROC["fx"]) * 4, (ka["y"] - np.float32(EUROC["cy"])) / np.float32(EUROC["fy"]) * 4,
                      np.full(n, 4, np.float32)], -1).astype(np.float32)
    normal = (world / np.linalg.norm(world, axis=1, keepdims=True)).astype(np.float32)
    mf = (np.linalg.norm(world, axis=1) * 1.5).astype(np.float32)
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    lsf = np.log(np.float32(1.2))
    na, nb = da[:, 0].astype(np.int32) >> 4, db[:, 0].astype(np.int32) >> 4
    ko, fo, ki, fi = [0], [0], [], []
    for nd in sorted(set(na.tolist()) & set(nb.tolist())):
        ki.extend(np.nonzero(na == nd)[0]); fi.extend(np.nonzero(nb == nd)[0])
        ko.append(len(ki)); fo.append(len(fi))
    off = np.concatenate([[0], np.cumsum(rng.integers(1, 40, 120))]).astype(np.int32)
    dd = da[np.arange(off[-1]) % n]
    A, B, _, bp = two_view_scene(5, w, h, EUROC)
    kA, _ = ex.extract(A)
    wp = bp(kA["x"], kA["y"])
    q = rotvec_to_quat((0.004, -0.006, 0.003))
    T7 = np.array([q[0], q[1], q[2], q[3], 0.03, -0.02, 0.015], np.float32)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    ex.image_cache_reserve(2, w, h)
    ex.image_cache_put(0, A)
    ex.image_cache_put(1, B)
    px0 = np.stack([kA["x"], kA["y"]], -1).astype(np.float32) + 1.5
    left, right, _, _ = stereo_scene(6, w, h)
    prev = np.stack([ka["x"], ka["y"]], -1).astype(np.float32)
    ref = None
    for rep in range(4 * int(__import__('os').environ.get('YGZF_REPEATS', '30'))):
        parts = []
        parts += ex.extract_dso(img, existing=ka[::7])[:2]
        parts += ex.search_by_projection_last(cam, kb, db, ka, world, da, I, z, I, z, 15.0)[:2]
        parts += ex.search_local_points(cam, kb, db, world, normal, mf * np.float32(1.2), mf * np.float32(0.2), mf, I, z, z, lsf, da, 3.0)[:2]
        parts += ex.search_for_initialization(cam, ka, da, kb, db, prev, 100, 0.9, True)[:2]
        parts += ex.search_by_bow(ko, ki, fo, fi, np.ones(n, np.uint8), ka, da, kb, db, 0.7, True)[:2]
        parts.append(ex.distinctive_descriptors_batch(off, dd))
        parts += ex.find_direct_projection_batch(cam, 1, T7, np.zeros(len(kA), np.int32), np.tile(ident, (len(kA), 1)), kA, wp, px0)
        kl, dl = ex.extract(left)
        kr, dr = ex.extract(right)
        parts += ex.compute_stereo_matches(left, right, kl, dl, kr, dr, 0.11, 47.9)
        dg = _digest(*parts)
        ref = ref or dg
        assert dg == ref, "single-frame paths differ in repetition %d" % rep


def test_two_contexts_sharing_an_image_are_run_to_run_identical():
    """The Tracking order of calls over two contexts and three streams, repeated: ComputePyramid with the extraction queued ahead (extractor
    context: its stream + its copy stream), the image cache filled device to device from it (cache context, ordered by events), SparseImgAlign
    on the slots, then the queued extraction collected -- alternating two images so that every buffer is overwritten each time."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    from orb_ygz_slam_amd.scene import two_view_scene
    w, h = 752, 480
    imgA, imgB, _, backproject = two_view_scene(21, w, h, EUROC, rotvec=(0.002, 0.001, -0.003), trans=(0.01, -0.02, 0.01))
    fe = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    cache = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    cache.image_cache_reserve(4, w, h)
    fe.set_extract_ahead(True)
    cam = make_camera(w, h)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    inv = fe.tables()["inv_scale"]
    keys = {}
    for name, img in (("A", imgA), ("B", imgB)):
        k, _ = fe.extract(img)
        keys[name] = (k, backproject(k["x"], k["y"]) if name == "A" else np.stack([k["x"], k["y"], np.ones(len(k))], -1).astype(np.float32))
    ref = {}
    for rep in range(int(__import__('os').environ.get('YGZF_REPEATS', '30')) * 2):
        cur, prev = ("A", "B") if rep % 2 == 0 else ("B", "A")
        img = imgA if cur == "A" else imgB
        pyr = fe.compute_pyramid(img)
        slot = rep % 2
        cache.image_cache_put_resident(slot, fe)
        parts = list(pyr)
        if rep > 0:
            k, world = keys[prev]
            r = cache.sia_run_cached(cam, 1 - slot, slot, k, world, ident, ident, inv, 7, 1)
            parts += [np.asarray(r[1]), np.asarray([r[0]])]
        kk, dd = fe.extract_resident(w, h)
        parts += [kk, dd]
        if rep < 2:
            continue
        dg = _digest(*parts)
        ref.setdefault(cur, dg)
        assert dg == ref[cur], "repetition %d (%s) differs" % (rep, cur)
