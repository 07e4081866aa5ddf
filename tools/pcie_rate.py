#!/usr/bin/env python3
"""tools/pcie_rate.py -- the PCIe-inclusive rate of the hot path (host frames in, keypoints + descriptors + matches out), which is
NOT bench.py's `value` (that one starts with the frames resident in HBM): ygzf_extract_batch_host + ygzf_match_batch_prev + fetch of
every frame's results, pageable host memory, one stream."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bench import WORKLOADS, make_frames  # noqa: E402
from orb_ygz_slam_amd import Extractor, make_camera  # noqa: E402


def main():
    import torch
    torch.cuda.init()            # before the library creates its own HIP context (as bench.py does)
    w, h, nl, sf, nf, ini, mn = WORKLOADS["euroc752x480_8lvl_1000feat"]
    B, steps = 256, 5
    frames = make_frames(B, w, h)
    ex = Extractor(nf, sf, nl, ini, mn, max_width=w, max_height=h, max_batch=B)
    cam = make_camera(w, h)

    def step(fetch):
        ex.extract_batch_host(frames)
        ex.match_batch_prev(cam, 15.0, True, True, True)
        if fetch:
            for f in range(B):
                ex.batch_fetch(f)
                ex.match_fetch(f)
        ex.sync()
    step(True)
    res = {}
    for name, fetch in (("pageable_h2d_only", False), ("pageable_h2d_and_per_frame_fetch", True)):
        t0 = time.perf_counter()
        for _ in range(steps):
            step(fetch)
        res[name] = round(B * steps / (time.perf_counter() - t0), 1)
    # page-locked host buffers (what a capture pipeline would hand over) + one-shot result fetch
    pin = torch.from_numpy(frames).pin_memory()
    stride = ex.max_keypoints(w, h)
    from orb_ygz_slam_amd.capi import KP_DTYPE
    ok = torch.empty((B, stride, KP_DTYPE.itemsize), dtype=torch.uint8).pin_memory()
    od = torch.empty((B, stride, 32), dtype=torch.uint8).pin_memory()
    on = torch.empty(B, dtype=torch.int32).pin_memory()
    out = (ok.numpy().view(KP_DTYPE).reshape(B, stride), od.numpy(), on.numpy())
    pf = pin.numpy()

    def step2():
        ex.extract_batch_host(pf)
        ex.match_batch_prev(cam, 15.0, True, True, True)
        ex.batch_fetch_all(B, out)
    step2()
    t0 = time.perf_counter()
    for _ in range(steps):
        step2()
    res["pinned_h2d_and_one_shot_fetch"] = round(B * steps / (time.perf_counter() - t0), 1)
    # two contexts (own streams), software-pipelined: while one batch is in its kernels the other one's frames go up / results come down
    exs = [ex, Extractor(nf, sf, nl, ini, mn, max_width=w, max_height=h, max_batch=B)]
    pins = [pf, torch.from_numpy(frames).pin_memory().numpy()]
    outs = [out]
    ok2 = torch.empty((B, stride, KP_DTYPE.itemsize), dtype=torch.uint8).pin_memory()
    od2 = torch.empty((B, stride, 32), dtype=torch.uint8).pin_memory()
    on2 = torch.empty(B, dtype=torch.int32).pin_memory()
    outs.append((ok2.numpy().view(KP_DTYPE).reshape(B, stride), od2.numpy(), on2.numpy()))

    def submit(i):
        exs[i].extract_batch_host(pins[i])
        exs[i].match_batch_prev(cam, 15.0, True, True, True)
    submit(0); submit(1); exs[0].batch_fetch_all(B, outs[0]); exs[1].batch_fetch_all(B, outs[1])
    n2 = 4 * steps
    t0 = time.perf_counter()
    submit(0)
    for it in range(1, n2):
        submit(it & 1)                                  # enqueue the next batch on the other stream ...
        exs[(it - 1) & 1].batch_fetch_all(B, outs[(it - 1) & 1])   # ... then wait for / download the previous one
    exs[(n2 - 1) & 1].batch_fetch_all(B, outs[(n2 - 1) & 1])
    res["pinned_two_contexts_pipelined"] = round(B * n2 / (time.perf_counter() - t0), 1)
    ref_k, ref_d = ex.batch_fetch(3)
    assert (out[0][3][:out[2][3]] == ref_k).all() and (out[1][3][:out[2][3]] == ref_d).all()
    print(json.dumps({"pcie_inclusive_frames_per_s": res, "batch": B, "note": "host frames in (H2D), all keypoints + descriptors out (D2H); first three: one stream, serial; last: two contexts software-pipelined"}))


if __name__ == "__main__":
    main()
