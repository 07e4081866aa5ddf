#!/usr/bin/env python3
"""tools/pmc_shapes_summary.py <dir> <tag> <shapes...> -- condense what tools/profile_shapes.sh collected:
  <tag>_<shape>_kernel_stats.csv   rocprofv3 --stats of the bench command of that workload
  <tag>_<shape>_pmc_hbm.csv        FETCH_SIZE / WRITE_SIZE per launch, raw and corrected (profiles/hbm_calibration.json, as tools/pmc_summary.py)
  traffic_shapes.json              {workload key of bench.py's other_workloads: {kernel: HBM bytes per launch}} -- merged into profiles/traffic.json"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_summary as P

KEYS = {"uhd": ("uhd3840x2160_12lvl_8000feat_stereo", "sub-batches of 64 frames = 32 stereo pairs"),
        "fhd": ("fhd1920x1080_8lvl_4000feat", "sub-batches of 128 frames"),
        "align": ("euroc752x480_8lvl_1000feat_align", "sub-batches of 256 frames")}


def main():
    out, tag, shapes = sys.argv[1], sys.argv[2], sys.argv[3:]
    merged = {}
    for s in shapes:
        key, what = KEYS[s]
        P.copy_stats(out, tag, s + "_stats", "%s_%s_kernel_stats.csv" % (tag, s))
        # hbm_table() wants <out>/<prefix>fetch and <out>/<prefix>write and writes <tag>_<prefix>pmc_hbm.csv
        merged[key] = P.hbm_table(out, tag, s + "_", "python bench.py --steps 2 --warmup 1 of %s (%s)" % (key, what))
    json.dump(merged, open(os.path.join(out, "traffic_shapes.json"), "w"), indent=1)
    for s in shapes:
        print(open(os.path.join(out, "%s_%s_pmc_hbm.csv" % (tag, s))).read())


if __name__ == "__main__":
    main()
