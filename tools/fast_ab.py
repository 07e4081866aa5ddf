#!/usr/bin/env python3
"""tools/fast_ab.py -- the two forms of the FAST cell loop side by side on the bench clip: per-kernel event timings (ygzf_profile_*) of an
isolated 256-frame extract (+ match) with k_fast_quads (register staging, geometry derived per wave) and k_fast_tab (per-cell records,
LDS-DMA staging), both threshold plans.  usage: python tools/fast_ab.py [frames=256] [reps=10]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_frames  # noqa: E402
from orb_ygz_slam_amd import Extractor, make_camera  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    w, h = 752, 480
    frames = make_frames(nf, w, h)
    cam = make_camera(w, h)
    d = torch.from_numpy(frames).cuda()
    out = {}
    ref = None
    for kern in (1, 2):
        for plan in (2, 1):
            ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=nf)
            ex.set_fast_kernel(kern)
            ex.set_fast_plan(plan)
            for _ in range(3):
                ex.extract_batch_device(d.data_ptr(), nf, w, h)
                ex.match_batch_prev(cam, 15.0, True, True, True)
            ex.sync()
            ex.profile_enable(True)
            ex.profile_reset()
            for _ in range(reps):
                ex.extract_batch_device(d.data_ptr(), nf, w, h)
                ex.match_batch_prev(cam, 15.0, True, True, True)
                ex.sync()
            pr = ex.profile_read()
            out["kernel%d_plan%d" % (kern, plan)] = {k: round(1000.0 * ms / max(n, 1), 1) for k, (ms, n) in pr.items() if n}
            got = [ex.batch_fetch(f) for f in (0, nf // 2, nf - 1)]
            if ref is None:
                ref = got
            for (k, dd), (k0, d0) in zip(got, ref):
                assert (k == k0).all() and (dd == d0).all(), "results differ between the FAST kernels / plans"
            ex.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
