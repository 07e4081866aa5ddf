#!/usr/bin/env python3
"""tools/merge_traffic.py <prof dir of tools/profile_round.sh> <dir of tools/profile_shapes.sh> <out json> -- one traffic.json for bench.py: HBM bytes per launch per
kernel for the default workload (profile_round.sh's traffic.json) and for the FHD / UHD-stereo / align workloads (profile_shapes.sh's traffic_shapes.json)."""
import json
import sys

base = json.load(open(sys.argv[1] + "/traffic.json"))
shapes = json.load(open(sys.argv[2] + "/traffic_shapes.json"))
base["_source"] = base.get("_source", "") + "; other workloads: tools/profile_shapes.sh (same passes, same correction), keyed as bench.py's other_workloads"
base.update(shapes)
json.dump(base, open(sys.argv[3], "w"), indent=1)
print("merged", sorted(k for k in base if not k.startswith("_") and k not in ("fetch_factor", "write_factor")))
