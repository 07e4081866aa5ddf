"""The product's multi-GPU split (ygzf_mgpu_*) on the one GPU the test box has: device slots mapped onto device 0 (every slot its own
context, stream, staging and host thread).  A sharded batch must return the bytes of the unsharded one in input order, units (frame pairs) are
never split over slots, a slot that gets no frame is harmless, a device that does not exist is an error -- and bench.py's in-process sharding
flag still produces a valid line."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _clip(n, w, h):
    frames = np.empty((n, h, w), np.uint8)
    for i in range(n):
        if i % 4 == 0:
            scene = synth_frame(700 + i // 4, w + 16, h + 16)
        frames[i] = scene[2 * (i % 4):2 * (i % 4) + h, 3 * (i % 4):3 * (i % 4) + w]
    return frames


@pytest.mark.parametrize("unit", [1, 2, 4])
def test_sharded_equals_unsharded(oracle, unit):
    from orb_ygz_slam_amd import MultiGpu, make_camera
    w, h, n = 640, 480, 24
    frames = _clip(n, w, h)
    cam = make_camera(w, h)
    one = MultiGpu([0], max_width=w, max_height=h, max_frames_per_device=n)
    ref = one.extract_match(frames, unit=unit, cam=cam)
    one.close()
    for slots in ([0, 0], [0, 0, 0], [0] * 5):
        mg = MultiGpu(slots, max_width=w, max_height=h, max_frames_per_device=n)
        assert mg.device_count() == len(slots)
        got = mg.extract_match(frames, unit=unit, cam=cam)
        for f in range(n):                                            # a unit's frames share a slot, consecutive units take consecutive slots
            assert mg.slot_of_frame(f, unit) == (f // unit) % len(slots)
        mg.close()
        for a, b, name in zip(ref, got, ("kps", "desc", "n_kp", "match", "nmatches")):
            assert np.array_equal(a, b), (slots, unit, name)
    k, d, c, m, nm = ref
    assert (c > 500).all()
    assert ((nm == -1) == (np.arange(n) % unit == 0)).all()           # first frame of a unit has no predecessor
    if unit > 1:
        assert (nm[np.arange(n) % unit != 0] > 100).all()
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)                         # and the contents are the extractor's
    for f in (0, n // 2 + 1, n - 1):
        ok, od = oex.extract(frames[f])
        assert c[f] == len(ok) and (k[f, :c[f]] == ok).all() and (d[f, :c[f]] == od).all()


def test_extract_only_and_idle_slots():
    from orb_ygz_slam_amd import MultiGpu
    w, h = 320, 240
    frames = _clip(3, w, h)
    mg = MultiGpu([0, 0, 0, 0, 0, 0], 500, 1.2, 4, 20, 7, max_width=w, max_height=h, max_frames_per_device=4)      # more slots than frames
    k, d, c, m, nm = mg.extract_match(frames)
    assert m is None and nm is None and (c > 50).all()
    k2, d2, c2, _, _ = mg.extract_match(frames[::-1].copy())             # a second call on the same handle
    assert np.array_equal(c2, c[::-1]) and np.array_equal(k2, k[::-1]) and np.array_equal(d2, d[::-1])
    mg.close()


def test_long_slot_runs_in_chunks():
    """A slot with more than 128 frames goes through in chunks (gather / device work / scatter overlapped): 300 frames on one slot and on two
    return the same bytes, pair by pair, and the chunk borders (frames 127 | 128, 255 | 256 of the slot) are ordinary pairs."""
    from orb_ygz_slam_amd import MultiGpu, make_camera
    w, h, n = 320, 240, 300
    frames = np.ascontiguousarray(np.concatenate([_clip(20, w, h)] * 15)[:n])
    cam = make_camera(w, h)
    one = MultiGpu([0], 500, 1.2, 4, 20, 7, max_width=w, max_height=h, max_frames_per_device=n)
    ref = one.extract_match(frames, unit=n, cam=cam)                     # one unit: every frame but the first has its predecessor
    one.close()
    k, d, c, m, nm = ref
    assert nm[0] == -1 and (nm[1:] >= 0).all() and (c > 50).all()
    for f in (1, 21, 127, 128, 129, 255, 256, 299):                       # a frame equals the one 20 before it: same keypoints, same match row
        if f >= 21:
            assert c[f] == c[f - 20] and np.array_equal(k[f, :c[f]], k[f - 20, :c[f]]) and np.array_equal(m[f, :c[f]], m[f - 20, :c[f]]) and nm[f] == nm[f - 20]
    two = MultiGpu([0, 0], 500, 1.2, 4, 20, 7, max_width=w, max_height=h, max_frames_per_device=n)
    got = two.extract_match(frames, unit=150, cam=cam)                  # two units of 150 frames, one per slot (each more than one chunk)
    two.close()
    k2, d2, c2, m2, nm2 = got
    assert np.array_equal(c2, c) and np.array_equal(k2, k) and np.array_equal(d2, d)
    assert nm2[0] == -1 and nm2[150] == -1
    keep = np.ones(n, bool); keep[[0, 150]] = False
    assert np.array_equal(nm2[keep], nm[keep]) and np.array_equal(m2[keep], m[keep])


@pytest.mark.parametrize("nslots", [8, 16])
def test_eight_and_sixteen_slots(oracle, nslots):
    """BASELINE.json's literal multi-GPU configurations, on the one GPU the box has: batch = 8 frames, one per device slot (unit 1); batch = 16 frames =
    8 stereo pairs, a pair per slot (unit 2, ygzf_mgpu_extract_stereo: both eyes' extraction + ComputeStereoMatches).  Same bytes as one slot doing
    everything, and as the oracle."""
    from orb_ygz_slam_amd import MultiGpu, Extractor, make_camera, EUROC
    w, h = 640, 480
    frames = _clip(16, w, h)
    cam = make_camera(w, h)
    one = MultiGpu([0], max_width=w, max_height=h, max_frames_per_device=16)
    ref8 = one.extract_match(frames[:8], unit=1, cam=cam)
    stereo = np.ascontiguousarray(frames.copy())
    for p in range(8):                                                   # right eye = the left image shifted by a disparity
        stereo[2 * p + 1, :, :w - 6 - p] = stereo[2 * p, :, 6 + p:]
    mb, mbf = 0.11, 47.9
    refs = one.extract_stereo(stereo, mb, mbf)
    one.close()
    mg = MultiGpu([0] * nslots, max_width=w, max_height=h, max_frames_per_device=4)
    got8 = mg.extract_match(frames[:8], unit=1, cam=cam)
    gots = mg.extract_stereo(stereo, mb, mbf)
    mg.close()
    for a, b, name in zip(ref8, got8, ("kps", "desc", "n_kp", "match", "nmatches")):
        assert np.array_equal(a, b), (nslots, name)
    for a, b, name in zip(refs, gots, ("kps", "desc", "n_kp", "u_right", "depth")):
        assert np.array_equal(a.view(np.uint8) if a.dtype == np.float32 else a, b.view(np.uint8) if b.dtype == np.float32 else b), (nslots, name)
    k, d, c, ur, dp = gots
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    for p in (0, 5):
        kl, dl = oex.extract(stereo[2 * p])
        kr, dr = oex.extract(stereo[2 * p + 1])
        our, odp = oex.compute_stereo_matches(stereo[2 * p], stereo[2 * p + 1], kl, dl, kr, dr, mb, mbf)
        n = len(kl)
        assert c[2 * p] == n and np.array_equal(ur[p, :n].view(np.uint32), our.view(np.uint32)) and np.array_equal(dp[p, :n].view(np.uint32), odp.view(np.uint32))
        assert (our >= 0).sum() > 100 and (ur[p, n:] == -1).all()


def test_page_locked_frames_skip_the_staging_copy():
    """Frames that already lie in page-locked host memory are copied to the device from where they lie (hipPointerGetAttributes decides); same bytes
    as from pageable memory, chunks alternating between a slot's two contexts included (YGZF_FORCE=mgpu_chunk=6 makes them small)."""
    import ctypes as C
    from orb_ygz_slam_amd import MultiGpu, make_camera
    w, h, n = 320, 240, 44
    frames = np.ascontiguousarray(np.concatenate([_clip(20, w, h)] * 3)[:n])
    cam = make_camera(w, h)
    from orb_ygz_slam_amd.capi import force_env
    old_force = os.environ.get("YGZF_FORCE")
    os.environ["YGZF_FORCE"] = force_env(mgpu_chunk=6)
    try:
        mg = MultiGpu([0, 0], 500, 1.2, 4, 20, 7, max_width=w, max_height=h, max_frames_per_device=n)
        assert mg.chunk_frames() == 6
        ref = mg.extract_match(frames, unit=2, cam=cam)
        L = mg.L                                       # (page-locked memory from the library itself: ctypes finding "libamdhip64.so" by name may hand
        L.ygzf_alloc_host.restype = C.c_void_p         #  back another copy of the runtime -- torch bundles one -- that has no device initialised)
        L.ygzf_alloc_host.argtypes = [C.c_int, C.c_size_t]
        L.ygzf_free_host.argtypes = [C.c_void_p]
        ptr = C.c_void_p(L.ygzf_alloc_host(0, frames.nbytes))
        assert ptr.value
        pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(frames.nbytes,)).reshape(frames.shape)
        pinned[:] = frames
        got = mg.extract_match(pinned, unit=2, cam=cam)
        ref4 = mg.extract_match(frames, unit=4, cam=cam)               # 4 does not divide 6: chunks of 4 frames
        got4 = mg.extract_match(pinned, unit=4, cam=cam)
        mg.close()
        del pinned
        L.ygzf_free_host(ptr)
    finally:
        if old_force is None:
            del os.environ["YGZF_FORCE"]
        else:
            os.environ["YGZF_FORCE"] = old_force
    for a, b, name in zip(ref, got, ("kps", "desc", "n_kp", "match", "nmatches")):
        assert np.array_equal(a, b), name
    for a, b, name in zip(ref4, got4, ("kps", "desc", "n_kp", "match", "nmatches")):
        assert np.array_equal(a, b), name
    big = MultiGpu([0], 500, 1.2, 4, 20, 7, max_width=w, max_height=h, max_frames_per_device=n)      # one slot, one chunk: the unchunked answer
    want = big.extract_match(frames, unit=2, cam=cam)
    big.close()
    for a, b, name in zip(want, got, ("kps", "desc", "n_kp", "match", "nmatches")):
        assert np.array_equal(a, b), name


def test_missing_device_is_an_error():
    from orb_ygz_slam_amd import MultiGpu, YgzfError
    import torch
    with pytest.raises(YgzfError):
        MultiGpu([0, torch.cuda.device_count()], max_width=320, max_height=240, max_frames_per_device=2)


def test_bench_in_process_devices_on_one_gpu():
    """bench.py --devices-in-process 2 --reuse-devices: two logical devices (threads, contexts, gate barriers) on the box's single GPU"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--devices-in-process", "2", "--reuse-devices", "--steps", "2", "--warmup", "1",
                          "--batch", "768", "--passes", "1", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["config"]["devices_reused"] is True and line["config"]["devices_per_process"] == 2 and line["n_gpus"] == 2
    assert line["value"] > 0 and line["steps"] == 2 and line["config"]["frames_per_gpu_per_step"] == 768


def test_bench_two_ranks_on_one_gpu():
    """The driver's N > 1 launch (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) with both ranks on the box's one GPU
    (`--share-gpu`: rendezvous over gloo instead of RCCL, which refuses two ranks on one device): every barrier, every max-over-ranks reduction and the
    per-rank extras of the REAL bench path (the end-to-end pipeline, the other workloads) run in two processes -- what cannot be tried here is RCCL itself."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29571",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--batch", "768", "--passes", "1", "--other-steps", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                              # rank 0 prints the one line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["config"]["valid_measurement"] is False and line["config"]["processes"] == 2
    assert line["value"] > 0 and line["value_end_to_end"] > 0 and line["cpu_baseline"] is None and line["mgpu_end_to_end"] is None
    ow = line["other_workloads"]
    assert set(ow) >= {"fhd1920x1080_8lvl_4000feat", "uhd3840x2160_12lvl_8000feat_stereo", "euroc752x480_8lvl_1000feat_align"}
    assert all(v["value"] > 0 for v in ow.values()) and ow["fhd1920x1080_8lvl_4000feat"]["value_end_to_end"] > 0
    assert "vga640x480_8lvl_1000feat_extract_only" in ow and ow["vga640x480_8lvl_1000feat_extract_only"]["matches_per_frame"] is None
    pr = line["per_rank"]                                                 # a straggler shows: every rank's own elapsed time, not only the maximum
    assert len(pr["elapsed_s"]) == 2 and pr["max_s"] >= pr["min_s"] > 0 and pr["slowest_rank"] in (0, 1)
    assert abs(line["timed_region_s"] - pr["max_s"]) < 1e-3
    assert line["roofline"]["value_end_to_end"] == line["value_end_to_end"] == line["config"]["value_end_to_end"]


def test_bench_two_ranks_on_one_gpu_uhd_stereo():
    """the same two-process control flow on the workload the 4K scaling curve uses (BASELINE.json configs[4]: 3840x2160 stereo pairs, 12 levels, 8000
    features): every rank its own clip of pairs, ComputeStereoMatches inside the step, barriers and max-over-ranks around it"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--workload", "uhd3840x2160_12lvl_8000feat", "--stereo", "--steps", "2", "--warmup", "1",
           "--sub-batch", "8", "--streams", "2", "--distinct", "8", "--no-extras"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["workload"] == "uhd3840x2160_12lvl_8000feat" and cfg["stereo"] is True and cfg["valid_measurement"] is False
    assert line["value"] > 0 and 7000 < line["keypoints_per_frame"] <= 8100
    assert len(line["per_rank"]["elapsed_s"]) == 2


def test_bench_falls_back_from_rccl_to_gloo_together():
    """`--backend auto`: the control plane is gloo, RCCL carries the timing barrier only when its communicator comes up on EVERY rank.  Two ranks on the
    box's one GPU make RCCL refuse (duplicate device): both ranks must agree on gloo over the control plane and the run must still produce its line,
    saying which transport it used and why."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29573",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--share-gpu-try-rccl", "--backend", "auto", "--steps", "2", "--warmup", "1",
           "--batch", "768", "--passes", "1", "--no-extras"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["value"] > 0 and cfg["processes"] == 2
    assert cfg["barrier_backend"] in ("gloo", "rccl")
    if cfg["barrier_backend"] == "gloo":
        assert cfg["barrier_backend_note"]                              # the reason travels with the line
    # and an explicit --backend gloo needs no RCCL at all
    cmd2 = [c for c in cmd if c != "--share-gpu-try-rccl"]
    cmd2[cmd2.index("auto")] = "gloo"
    cmd2[cmd2.index("29573")] = "29574"
    out = subprocess.run(cmd2, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert line["config"]["barrier_backend"] == "gloo" and line["value"] > 0


def test_small_calls_take_the_packed_result_path(oracle):
    """One frame / one pair per call (BASELINE.json's literal multi-GPU configurations; a Tracking frame): the slot brings the results over in ONE copy
    (ygzf_batch_fetch_packed) -- also for frames smaller than the handle's maximum, whose keypoint rows are shorter than the handle's stride."""
    from orb_ygz_slam_amd import MultiGpu, make_camera
    mg = MultiGpu([0], max_width=752, max_height=480, max_frames_per_device=2)
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    for (w, h) in ((752, 480), (640, 400), (320, 240)):
        frames = _clip(2, w, h)
        k, d, c, m, nm = mg.extract_match(frames, unit=2, cam=make_camera(w, h))
        for f in range(2):
            ok, od = oex.extract(frames[f])
            assert c[f] == len(ok) and (k[f, :c[f]] == ok).all() and (d[f, :c[f]] == od).all(), (w, h, f)
        assert nm[0] == -1 and nm[1] > 50
        k1, d1, c1, _, _ = mg.extract_match(frames[:1])                   # one frame, extraction only
        assert c1[0] == c[0] and (k1[0, :c1[0]] == k[0, :c[0]]).all() and (d1[0, :c1[0]] == d[0, :c[0]]).all()
    mg.close()


def test_one_pair_per_slot_takes_two_streams(oracle):
    """A slot that is handed exactly ONE (left, right) pair in page-locked memory puts the eyes on its two contexts -- the right eye's upload beside the
    left eye's kernels, ComputeStereoMatches across the two contexts (ygzf_stereo_pair_host) -- and must return the bytes of the one-context path
    (the same pair from pageable memory) and of the oracle: keypoints, descriptors of both eyes, mvuRight / mvDepth."""
    import ctypes as C
    from orb_ygz_slam_amd import MultiGpu
    mb, mbf = 0.11, 47.9
    mg = MultiGpu([0], max_width=752, max_height=480, max_frames_per_device=2)
    L = mg.L
    L.ygzf_alloc_host.restype = C.c_void_p
    L.ygzf_alloc_host.argtypes = [C.c_int, C.c_size_t]
    L.ygzf_free_host.argtypes = [C.c_void_p]
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    for (w, h, dsp) in ((752, 480, 9), (640, 400, 14), (752, 480, 3)):
        left = _clip(1, w, h)[0]
        pair = np.stack([left, np.zeros_like(left)])
        pair[1, :, :w - dsp] = left[:, dsp:]
        ref = mg.extract_stereo(np.ascontiguousarray(pair), mb, mbf)                      # pageable: one context, both eyes in one launch
        ptr = C.c_void_p(L.ygzf_alloc_host(0, pair.nbytes))
        assert ptr.value
        pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(pair.nbytes,)).reshape(pair.shape)
        pinned[:] = pair
        for _ in range(2):                                                                # twice: the contexts' carry / reuse paths
            got = mg.extract_stereo(pinned, mb, mbf)
            for a, b, name in zip(ref, got, ("kps", "desc", "n_kp", "u_right", "depth")):
                assert np.array_equal(a.view(np.uint8) if a.dtype == np.float32 else a, b.view(np.uint8) if b.dtype == np.float32 else b), (w, h, name)
        del pinned
        L.ygzf_free_host(ptr)
        k, d, c, ur, dp = ref
        kl, dl = oex.extract(pair[0])
        kr, dr = oex.extract(pair[1])
        our, odp = oex.compute_stereo_matches(pair[0], pair[1], kl, dl, kr, dr, mb, mbf)
        n = len(kl)
        assert c[0] == n and c[1] == len(kr) and (k[1, :c[1]] == kr).all() and (d[1, :c[1]] == dr).all()
        assert np.array_equal(ur[0, :n].view(np.uint32), our.view(np.uint32)) and np.array_equal(dp[0, :n].view(np.uint32), odp.view(np.uint32))
        assert (our >= 0).sum() > 100
    mg.close()
