#!/usr/bin/env python3
"""tools/partition_sweep.py -- sweep of ygzf_set_stream_partition on the bench's own resident pipeline (752x480 / 8 levels / 1000 features, 256-frame
sub-batches rotating over S contexts): for every (contexts, fill_cus, main_mode) the resident rate, and every kernel's in-pipeline average against its
isolated average (the "stretch").  One JSON line per setting + a summary table on stderr.

    python tools/partition_sweep.py [--streams 3,4] [--fills 0,-1,32,64,96,128] [--complement 0,1] [--steps 4]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="3,4")
    ap.add_argument("--fills", default="0,-1,32,64,96,128")
    ap.add_argument("--complement", default="0,1")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=14)
    ap.add_argument("--workload", default="euroc752x480_8lvl_1000feat")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    wl = args.workload
    sub, _ = bench.SHAPES[wl]
    rows = []
    for S in [int(x) for x in args.streams.split(",")]:
        rounds = max(1, (args.rounds * 3) // S)
        pipe = bench.Pipeline(0, wl, sub, rounds, S, 1000, passes=2, distinct=S * sub if "euroc" in wl else 24)
        pipe.step(); pipe.sync()
        iso = None
        for fill in [int(x) for x in args.fills.split(",")]:
            for mm in [int(x) for x in args.complement.split(",")]:
                if mm == 1 and fill <= 0:
                    continue
                for e in pipe.exs:
                    e.set_stream_partition(fill, mm)
                if iso is None:
                    iso = bench.isolated_pass(pipe)
                pipe.step(); pipe.sync()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    pipe.step()
                pipe.sync()
                el = time.perf_counter() - t0
                fps = pipe.batch * pipe.passes * args.steps / el
                # a profiled step afterwards (events perturb little, but the rate above is the unprofiled one)
                pipe.profile(True)
                pipe.step(); pipe.sync()
                kt = bench.kernel_table(pipe.profile_read())
                pipe.profile(False)
                stretch = {k: round(v["avg_us"] / iso[k], 2) for k, v in kt.items() if iso.get(k)}
                row = {"contexts": S, "fill_cus": fill, "main_mode": mm, "frames_per_s": round(fps, 1), "kernels_us": {k: v["avg_us"] for k, v in kt.items()},
                       "isolated_us": iso, "stretch": stretch}
                rows.append(row)
                print(json.dumps(row), flush=True)
        for e in pipe.exs:
            e.close()
        del pipe
        torch.cuda.empty_cache()
    print("contexts fill main   frames/s   stretch(fast,describe,octree,match,pyr)", file=sys.stderr)
    for r in rows:
        st = r["stretch"]
        print("%8d %4d %4d %10.0f   %s" % (r["contexts"], r["fill_cus"], r["main_mode"], r["frames_per_s"],
                                           " ".join("%.2f" % st.get(k, 0) for k in ("k_fast_tab", "k_describe", "k_octree", "k_match_last", "k_pyr_resize"))), file=sys.stderr)


if __name__ == "__main__":
    main()
