#!/usr/bin/env python3
"""tools/mgpu_ab.py -- ygzf_mgpu_extract_match's rate (bench.mgpu_end_to_end) beside the bare two-context pipeline (bench.end_to_end) on the same box:
    YGZF_MGPU_ORDER=0|1 python tools/mgpu_ab.py [frames_per_slot]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    from orb_ygz_slam_amd.capi import bind_host_thread_to_device
    cfg = bench.WORKLOADS["euroc752x480_8lvl_1000feat"]
    fr = bench.make_frames(768, 752, 480)
    cpus = bind_host_thread_to_device(0)
    out = {"order": os.environ.get("YGZF_MGPU_ORDER", "1"), "numa_cpus": cpus}
    for rep in range(2):
        r = bench.mgpu_end_to_end([0], cfg, fr)
        out["mgpu_%d" % rep] = (r["value"], r["value_pageable"])
    pipe = bench.Pipeline(0, "euroc752x480_8lvl_1000feat", 256, 1, 2, 1000, frames=fr)
    pipe.step(); pipe.sync()
    n, sec, link = bench.end_to_end(pipe, min_seconds=1.0)
    out["two_context_pipeline"] = round(n / sec, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
