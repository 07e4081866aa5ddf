"""tools/mgpu_rate.py -- ygzf_mgpu_extract_match on the box's GPU(s): frames/s from pageable and from page-locked host frames (bench.py's mgpu_end_to_end),
and BASELINE's literal batch-8 / batch-16 configurations (mgpu_literal_configs)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
devices = list(range(max(1, torch.cuda.device_count())))
cfg = bench.WORKLOADS["euroc752x480_8lvl_1000feat"]
frames = bench.make_frames(256, 752, 480, seed0=1000)
print(json.dumps(bench.mgpu_end_to_end(devices, cfg, frames)["runs"]))
if "--literal" in sys.argv:
    print(json.dumps(bench.mgpu_literal_configs(devices)))
