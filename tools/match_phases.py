#!/usr/bin/env python3
"""tools/match_phases.py -- phase clocks of k_match_last (SearchByProjection(Cur, Last)) for the first pair of a 256-frame batch
(YGZF_MATCH_DEBUG: the switch is read when the context is created)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YGZF_MATCH_DEBUG"] = "1"
import bench  # noqa: E402
from orb_ygz_slam_amd import Extractor, make_camera  # noqa: E402

frames = bench.make_frames(256, 752, 480, seed0=1000)
ex = Extractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=256)
cam = make_camera(752, 480)
for _ in range(2):
    ex.extract_batch_host(frames)
    ex.match_batch_prev(cam, 15.0, True, True, True)
    ex.sync()
