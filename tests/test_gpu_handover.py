"""The matcher's split path (several workgroups per pair, hand-over through device-scope accesses: csrc/match_kernels.hip) under load: one context
runs few-pair launches (split = 64 workgroups per pair) hundreds of times while two other contexts keep the chip busy with the 256-frame
pipeline on their own streams; every repetition must return the bytes of (a) the first one and (b) a context that takes the one-wave serial pass
without any split (YGZF_FORCE=match_serial=1,match_split=1) and (c) one that keeps the full fences (match_fence=1).  The reference:
ORBmatcher::SearchByProjection(Cur, Last) src/ORBmatcher.cc:1218-1350 (sequential ownership)."""
import hashlib
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(force, **kw):
    """the library reads YGZF_FORCE (plan pins, csrc/ygzf_internal.h) when a context is created"""
    from orb_ygz_slam_amd import Extractor
    from orb_ygz_slam_amd.capi import force_env
    old = os.environ.get("YGZF_FORCE")
    os.environ["YGZF_FORCE"] = force_env(**force)
    try:
        return Extractor(1000, 1.2, 8, 20, 7, **kw)
    finally:
        if old is None:
            os.environ.pop("YGZF_FORCE", None)
        else:
            os.environ["YGZF_FORCE"] = old


def test_split_matcher_under_load_equals_serial_pass():
    from bench import make_frames
    from orb_ygz_slam_amd import make_camera, EUROC
    w, h = 752, 480
    reps = int(os.environ.get("YGZF_HANDOVER_REPS", "200"))
    clip = make_frames(256, w, h, seed0=4242)
    cam = make_camera(w, h)
    small = clip[8:13]                                            # 5 frames -> 5 pairs per launch (the first against the carried frame): split path
    split = _ctx({}, max_width=w, max_height=h, max_batch=5)
    serial = _ctx({"match_serial": 1, "match_split": 1}, max_width=w, max_height=h, max_batch=5)
    fenced = _ctx({"match_fence": 1}, max_width=w, max_height=h, max_batch=5)
    loaders = [_ctx({}, max_width=w, max_height=h, max_batch=256) for _ in range(2)]

    # one-pair form too: SearchByProjection(cur, last) on host arrays
    ka, da = split.extract(clip[8])
    kb, db = split.extract(clip[9])
    world = np.stack([(ka["x"] - np.float32(EUROC["cx"])) / np.float32(EUROC["fx"]), (ka["y"] - np.float32(EUROC["cy"])) / np.float32(EUROC["fy"]),
                      np.ones(len(ka), np.float32)], -1).astype(np.float32)
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)

    def one(ex):
        hsh = hashlib.sha256()
        ex.extract_batch_host(small)
        ex.match_batch_prev(cam, 15.0, True, True, True)
        hsh.update(ex.match_counts()[1:].tobytes())
        for f in range(1, 5):
            m, o = ex.match_fetch(f)
            hsh.update(m.tobytes()); hsh.update(o.tobytes())
        n, m, o = ex.search_by_projection_last(cam, kb, db, ka, world, da, I, z, I, z, 15.0)[:3]
        hsh.update(np.asarray([n]).tobytes()); hsh.update(np.asarray(m).tobytes()); hsh.update(np.asarray(o).tobytes())
        return hsh.hexdigest(), n

    want, n1 = one(serial)
    assert n1 > 100
    assert one(fenced)[0] == want
    stop = threading.Event()
    launched = [0, 0]

    def load(i):
        e = loaders[i]
        while not stop.is_set():
            e.extract_batch_host(clip)
            e.match_batch_prev(cam, 15.0, True, True, True)
            e.sync()
            launched[i] += 1
    th = [threading.Thread(target=load, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    try:
        bad = []
        for r in range(reps):
            got = one(split)[0]
            if got != want:
                bad.append(r)
        assert not bad, "split hand-over differs from the serial pass in repetitions %s of %d" % (bad[:10], reps)
        assert one(fenced)[0] == want and one(serial)[0] == want
    finally:
        stop.set()
        for t in th:
            t.join()
    assert min(launched) >= 2, "the load contexts did not run beside the repetitions (%s)" % launched
    for e in [split, serial, fenced] + loaders:
        e.close()


def test_octree_helper_workgroups_under_load():
    """k_octree's helper workgroups (launches of a few frames, histogram plan: csrc/extract_kernels.hip) hand their share of a level's keys to
    workgroup 0 through a release / acquire pair executed by one thread per workgroup: one 1920x1080 frame and three 752x480 frames with forced
    helpers, hundreds of times, while two other contexts keep the chip busy with the 256-frame pipeline -- every repetition must return the bytes
    of a context without helpers (YGZF_FORCE=oct_helpers=1).  The reference: ORBextractor::DistributeOctTree, src/ORBextractor.cc:533-723."""
    from bench import make_frames
    from orb_ygz_slam_amd import Extractor, make_camera
    from orb_ygz_slam_amd.capi import force_env
    reps = int(os.environ.get("YGZF_HANDOVER_REPS", "200"))

    def ctx(force, nf, w, h, mb):
        old = os.environ.get("YGZF_FORCE")
        os.environ["YGZF_FORCE"] = force_env(**force)
        try:
            return Extractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=mb)
        finally:
            if old is None:
                os.environ.pop("YGZF_FORCE", None)
            else:
                os.environ["YGZF_FORCE"] = old

    big = make_frames(1, 1920, 1080, seed0=991)
    small = make_frames(3, 752, 480, seed0=992)
    clip = make_frames(256, 752, 480, seed0=4243)
    cam = make_camera(752, 480)
    cases = [(big, ctx({}, 4000, 1920, 1080, 1), ctx({"oct_helpers": 1}, 4000, 1920, 1080, 1)),
             (small, ctx({"oct_plan": "hist", "oct_helpers": 8}, 1000, 752, 480, 3), ctx({"oct_plan": "hist", "oct_helpers": 1}, 1000, 752, 480, 3))]
    loaders = [ctx({}, 1000, 752, 480, 256) for _ in range(2)]

    def one(ex, frames):
        hsh = hashlib.sha256()
        ex.extract_batch_host(frames)
        for f in range(len(frames)):
            k, d = ex.batch_fetch(f)
            hsh.update(np.ascontiguousarray(k).tobytes()); hsh.update(np.ascontiguousarray(d).tobytes())
        return hsh.hexdigest()

    want = [one(plain, frames) for frames, _, plain in cases]
    stop = threading.Event()
    launched = [0, 0]

    def load(i):
        e = loaders[i]
        while not stop.is_set():
            e.extract_batch_host(clip)
            e.match_batch_prev(cam, 15.0, True, True, True)
            e.sync()
            launched[i] += 1
    th = [threading.Thread(target=load, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    try:
        bad = []
        for r in range(reps):
            for ci, (frames, helped, _) in enumerate(cases):
                if one(helped, frames) != want[ci]:
                    bad.append((r, ci))
        assert not bad, "helper hand-over differs from the single-workgroup octree in (repetition, case) %s of %d" % (bad[:10], reps)
    finally:
        stop.set()
        for t in th:
            t.join()
    assert min(launched) >= 2, "the load contexts did not run beside the repetitions (%s)" % launched
    for frames, a, b in cases:
        a.close(); b.close()
    for e in loaders:
        e.close()
