#!/usr/bin/env python3
"""tools/e2e_ab.py -- SURVEY 8(d)'s transfers-included rate (bench.end_to_end) for one host layout of the frames:
    python tools/e2e_ab.py <workload> <host_pitch: 0 = tight rows | -1 = ygzf_host_row_pitch> [depth] [seconds]
The shape of the upload itself is the library's (YGZF_UPLOAD_K: 0 whole frames, 1 image rows, k runs of k rows)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "euroc752x480_8lvl_1000feat"
    hp = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    secs = float(sys.argv[4]) if len(sys.argv) > 4 else 1.5
    import torch
    torch.cuda.init()
    w = bench.WORKLOADS[wl][0]
    sub, _ = bench.SHAPES[wl]
    pipe = bench.Pipeline(0, wl, sub, 1, 2, 1000, distinct=min(2 * sub, 8 if "uhd" in wl else 24 if "fhd" in wl else 2 * sub))
    pipe.step(); pipe.sync()
    n, sec, link = bench.end_to_end(pipe, min_seconds=secs, depth=depth, host_pitch=(w if hp == 0 else None if hp < 0 else hp))
    print(json.dumps({"workload": wl, "host_pitch": link["host_row_pitch"], "upload_k": os.environ.get("YGZF_UPLOAD_K", "0"), "depth": depth,
                      "frames_per_s": round(n / sec, 1), "up_GBs": round(link["up"] * n / sec / 1e9, 2),
                      "fill_cus": os.environ.get("YGZF_FILL_CUS")}))


if __name__ == "__main__":
    main()
