#!/usr/bin/env python3
"""bench.py -- frames/s of the ORB extract + match hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (ORBextractor::operator() on every frame of a batch + ORBmatcher::SearchByProjection
of every frame against its predecessor) over one batch of synthetic frames that is already resident in HBM when the
timed region starts.  One process per GPU; frames are independent so each rank owns its own batch (weak scaling) and
there is no collective in the data path -- torch.distributed is used only for the barrier and the max-over-ranks time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 20 --warmup 3

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the definition of every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (w, h, nlevels, scale_factor, nfeatures, iniTh, minTh)
    "euroc752x480_8lvl_1000feat": (752, 480, 8, 1.2, 1000, 20, 7),      # BASELINE.json metric / configs[2] shape
    "vga640x480_8lvl_1000feat": (640, 480, 8, 1.2, 1000, 20, 7),        # configs[1]
    "fhd1920x1080_8lvl_4000feat": (1920, 1080, 8, 1.2, 4000, 20, 7),    # configs[3]
    "uhd3840x2160_12lvl_8000feat": (3840, 2160, 12, 1.2, 8000, 20, 7),  # configs[4] (one eye)
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def level_sizes(w, h, nlevels, sf):
    """Pyramid level sizes with the reference's float arithmetic (src/ORBextractor.cc:419-428, 1131-1132)."""
    scale = np.float32(1.0)
    out = []
    for l in range(nlevels):
        if l > 0:
            scale = np.float32(scale * np.float32(sf))
        inv = np.float32(1.0) / scale
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def algorithmic_bytes(w, h, nlevels, sf, nfeat):
    """algorithmic bytes per frame, total and per kernel (see DESIGN.md for the attribution)."""
    sz = level_sizes(w, h, nlevels, sf)
    P = sum(a * b for a, b in sz)
    P0, Pl = sz[0][0] * sz[0][1], sz[-1][0] * sz[-1][1]
    K = nfeat
    per = {
        "k_pyr_resize": (P - Pl) + (P - P0),              # read levels 0..L-2, write levels 1..L-1
        "k_fast_quads": P,                                 # FAST read of every level
        "k_describe": 2 * P + K * (749 + 512 + 32 + 28),   # blur read+write, orientation disc, samples, descriptor, KeyPoint
        "k_match_last": (K + K) * 32 + K * 8,              # B_match
        "k_octree": 0,                                     # candidate lists only (not part of SURVEY's formula)
    }
    total = per["k_pyr_resize"] + per["k_fast_quads"] + per["k_describe"] + per["k_match_last"]
    return total, per


def make_frames(n, w, h, seed0=1000):
    """Synthetic clip: groups of 8 consecutive frames are shifted crops of one scene so that frame-to-frame matching
    has something to find; every 8th frame is a scene cut."""
    from orb_ygz_slam_amd.synth import synth_frame
    frames = np.empty((n, h, w), np.uint8)
    m = 24
    scene = None
    for i in range(n):
        if i % 8 == 0:
            scene = synth_frame(seed0 + i // 8, w + m, h + m)
        dx, dy = (3 * (i % 8)) % m, (2 * (i % 8)) % m
        frames[i] = scene[dy:dy + h, dx:dx + w]
    return frames


def effective_cores():
    """Cores this process may actually use: affinity mask, further limited by a cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(frames, cfg, seconds_budget=12.0):
    """The CPU oracle ('port' of the reference path; the reference itself cannot be built here) on this host's cores:
    every worker thread runs extract + projection match over its own run of consecutive frames of the same clip for a
    bounded time (about `seconds_budget` s), so the default bench run stays within minutes on any host."""
    from oracle import oracle_py as O
    w, h, nl, sf, nf, ini, mn = cfg
    cores = effective_cores()
    sec1, _, _, n1 = O.bench_extract_match(frames, nf, sf, nl, ini, mn, threads=1, frames_per_thread=1000, max_seconds=2.0)
    fps1 = n1 / max(sec1, 1e-6)
    sec, nk, nm, n = O.bench_extract_match(frames, nf, sf, nl, ini, mn, threads=cores, frames_per_thread=100000,
                                           max_seconds=seconds_budget)
    n = max(n, 1)
    return {"value": round(n / sec, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d threads, each on its own run of consecutive frames of the bench clip, time-bounded: %d frames in %.1f s; "
                      "1 thread: %.2f frames/s" % (cores, n, sec, fps1),
            "keypoints_per_frame": round(nk / n, 1), "matches_per_frame": round(nm / n, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=768, help="frames per GPU per step")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent extractor contexts (own HIP stream + buffers) the batch is split over, so that the "
                         "latency-bound kernels of one sub-batch overlap the throughput-bound kernels of another")
    ap.add_argument("--workload", default="euroc752x480_8lvl_1000feat", choices=sorted(WORKLOADS))
    ap.add_argument("--align", action="store_true",
                    help="BASELINE config 3: also run SparseImgAlign (levels L-1..1, 10 iterations) of every frame against its "
                         "predecessor; `metric` stays extract+match, the step simply carries the extra work (see config.align)")
    ap.add_argument("--stereo", action="store_true",
                    help="BASELINE config 5 shape: the batch holds (left, right) pairs; Frame::ComputeStereoMatches runs on every pair "
                         "after extraction (extra work inside the step, see config.stereo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--plumbing-selftest", action="store_true",
                    help="CPU-only check of the multi-process plumbing (gloo): no GPU work, output is NOT a measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    cfg = WORKLOADS[args.workload]
    w, h, nl, sf, nf, ini, mn = cfg
    B = args.batch

    import torch
    dist = None
    use_gpu = not args.plumbing_selftest
    if use_gpu and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)          # before the process group: RCCL binds its communicator to the current device
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl" if use_gpu else "gloo")

    def barrier():
        if dist is not None:
            if use_gpu:
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    if args.plumbing_selftest:
        # exercises sharding, barrier and max-over-ranks reduction without a GPU; never a valid measurement
        barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"plumbing_selftest": True, "n_gpus": world, "frames_per_rank": B, "max_elapsed_s": float(el[0]),
                              "valid_measurement": False}))
        if dist is not None:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    from orb_ygz_slam_amd import Extractor, make_camera

    frames = make_frames(B, w, h, seed0=1000 + 97 * rank)   # every rank owns its own clip (one-frame-per-GPU sharding at scale)
    d_frames = torch.from_numpy(frames).to("cuda:%d" % local_rank)
    S = max(1, args.streams)
    if B % S or (args.stereo and (B // S) % 2):
        raise SystemExit("--batch must be a multiple of --streams (and of 2 x --streams with --stereo)")
    Bs = B // S
    exs = [Extractor(nf, sf, nl, ini, mn, max_width=w, max_height=h, max_batch=Bs, device=local_rank) for _ in range(S)]
    ex = exs[0]
    cam = make_camera(w, h)
    ptrs = [d_frames.data_ptr() + i * Bs * w * h for i in range(S)]

    def step():
        for e, ptr in zip(exs, ptrs):
            e.extract_batch_device(ptr, Bs, w, h)
            e.match_batch_prev(cam, 15.0, True, True, True)
            if args.align:
                e.align_batch_prev(cam, nl - 1, 1, 10)
            if args.stereo:
                e.stereo_batch(0.11, 47.9)

    for _ in range(args.warmup):
        step()
    for e in exs:
        e.sync()
    if not args.no_profile:
        for e in exs:
            e.profile_enable(True)
            e.profile_reset()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for e in exs:
        e.sync()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda:%d" % local_rank)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el[0])

    prof = {}
    if not args.no_profile:
        for e in exs:
            for name, (ms, n) in e.profile_read().items():
                a = prof.get(name, (0.0, 0))
                prof[name] = (a[0] + ms, a[1] + n)
            e.profile_enable(False)
    # untimed extra pass: the same sub-batches one context at a time, so that per-kernel durations are not stretched by the other streams
    iso = {}
    if not args.no_profile:
        for e, ptr in zip(exs, ptrs):
            e.profile_enable(True)
            e.profile_reset()
            for _ in range(2):
                e.extract_batch_device(ptr, Bs, w, h)
                e.match_batch_prev(cam, 15.0, True, True, True)
                e.sync()
            for name, (ms, n) in e.profile_read().items():
                a = iso.get(name, (0.0, 0))
                iso[name] = (a[0] + ms, a[1] + n)
            e.profile_enable(False)
    kp_counts = np.concatenate([e.batch_counts() for e in exs])
    m_counts = np.concatenate([e.match_counts() for e in exs])

    if rank == 0:
        total_frames = world * B * args.steps
        fps = total_frames / elapsed
        total_bytes, per_kernel = algorithmic_bytes(w, h, nl, sf, nf)
        roofline = None
        kernels = {}
        if prof:
            for name, (ms, n) in prof.items():
                if n:
                    kernels[name] = {"launches": n, "avg_us": round(1e3 * ms / n, 2), "total_ms": round(ms, 3)}
            # dominant = largest total time among the kernels SURVEY's byte formula covers
            cand = [k for k in kernels if per_kernel.get(k, 0) > 0]
            dom = max(cand, key=lambda k: kernels[k]["total_ms"])
            launches_per_step = kernels[dom]["launches"] / args.steps
            bytes_per_launch = per_kernel[dom] * B / launches_per_step   # pyramid: 7 launches share its bytes; S sub-batches
            avg_s = kernels[dom]["total_ms"] / kernels[dom]["launches"] * 1e-3
            achieved = bytes_per_launch / avg_s / 1e9
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile):
                try:
                    traffic = json.load(open(tfile)).get(args.workload, {}).get(dom)
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "algorithmic_bytes_per_launch": int(bytes_per_launch),
                        "isolated_avg_us": round(1e3 * iso[dom][0] / iso[dom][1], 2) if iso.get(dom, (0, 0))[1] else None,
                        "isolated_frac": round(bytes_per_launch / (iso[dom][0] / iso[dom][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                        if iso.get(dom, (0, 0))[1] else None,
                        "pipeline_achieved": round(total_bytes * fps / world / 1e9, 2),
                        "pipeline_frac": round(total_bytes * fps / world / 1e9 / HBM_PEAK_GBS, 5)}
        out = {
            "metric": "frames/s ORB extract+match, 752x480 8-lvl 1000-feat; 1->8 GPU scaling",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": args.workload, "width": w, "height": h, "levels": nl, "scale_factor": sf, "features": nf,
                       "frames_per_gpu_per_step": B, "streams": S, "align": bool(args.align), "stereo": bool(args.stereo), "match": "SearchByProjection(cur,last) th=15, identity pose",
                       "sharding": "one clip per GPU, no collective"},
            "keypoints_per_frame": round(float(kp_counts.mean()), 1), "matches_per_frame": round(float(m_counts.mean()), 1),
            "roofline": roofline, "kernels": kernels,
            "kernels_isolated_avg_us": {k: round(1e3 * v[0] / v[1], 2) for k, v in iso.items() if v[1]},   # one stream at a time (untimed pass)
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, cfg, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
