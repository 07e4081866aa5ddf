#!/usr/bin/env python3
"""tools/stress.py -- context churn (create / use / destroy) and repeated batches on one context; prints device memory in use
(rocm-smi) to spot leaks.  Run on the GPU box."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_ygz_slam_amd import Extractor, make_camera
from orb_ygz_slam_amd.synth import synth_frame


def vram():
    try:
        d = json.loads(subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--json"], capture_output=True, text=True).stdout)
        c = list(d.values())[0]
        return round(int([v for k, v in c.items() if "Used" in k][0]) / 1e6, 1)
    except Exception:
        return -1


img = synth_frame(0, 752, 480)
cam = make_camera(752, 480)
print("vram MB at start", vram())
for i in range(400):
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=4)
    ex.extract_batch_host(np.stack([img] * 4)); ex.match_batch_prev(cam, 15.0, True, True, True); ex.sync()
    ex.close()
    if i % 100 == 99:
        print("after", i + 1, "create/use/destroy cycles: vram MB", vram())
ex = Extractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=64)
frames = np.stack([synth_frame(i % 8, 752, 480) for i in range(64)])
ex.extract_batch_host(frames); ex.match_batch_prev(cam, 15.0, True, True, True)
ref = (ex.batch_counts().copy(), ex.match_counts().copy())
for it in range(300):
    ex.extract_batch_host(frames); ex.match_batch_prev(cam, 15.0, True, True, True)
ex.sync()
print("300 repeated batches stable:", bool((ex.batch_counts() == ref[0]).all() and (ex.match_counts()[1:] == ref[1][1:]).all()), "vram MB", vram())
