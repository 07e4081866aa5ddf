#!/usr/bin/env python3
"""tools/fast_pretest_ab.py -- k_fast_tab with / without the two-phase corner test on the synthetic bench clip and on the clip cut from the real image the
reference ships: isolated kernel time per 256-frame launch, resident rate of the three-context pipeline, and the sampled statistics
(corner-bearing quads and pre-test survivors per pass-1 run)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    wl = "euroc752x480_8lvl_1000feat"
    w, h = 752, 480
    clips = {"synthetic": bench.make_frames(768, w, h), "test1png": bench.make_frames_test1png(96, w, h)}
    for name, frames in clips.items():
        pipe = bench.Pipeline(0, wl, 256, 3, 3, 1000, frames=frames, passes=2)
        for mode in (1, 2, 0):
            for e in pipe.exs:
                e.set_fast_pretest(mode)
            for _ in range(3):
                pipe.step()
            pipe.sync()
            iso = bench.isolated_pass(pipe)
            t0 = time.perf_counter()
            for _ in range(6):
                pipe.step()
            pipe.sync()
            el = time.perf_counter() - t0
            st = pipe.exs[0].fast_stats()
            print(json.dumps({"clip": name, "pretest": {0: "auto", 1: "never", 2: "always"}[mode], "frames_per_s": round(pipe.batch * pipe.passes * 6 / el, 1),
                              "k_fast_tab_isolated_us": iso.get("k_fast_tab"), "chosen_on": st[0], "corner_quads_per_run": round(st[1], 1),
                              "survivors_per_run": round(st[2], 1), "fast_plan": pipe.exs[0].fast_plan()}), flush=True)
        for e in pipe.exs:
            e.close()
        del pipe
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
