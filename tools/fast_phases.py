"""tools/fast_phases.py [workload ...] -- where a wave of k_fast_tab and of k_describe spends its life, phase by phase.

Needs the instrumented build (python -m orb_ygz_slam_amd.build --phase-clock -> orb_ygz_slam_amd/lib_ab/libygzf_clk.so; the product library carries
no stamps): a wave stamps s_memtime at its phase borders (after draining its outstanding memory / LDS operations, so a phase is charged with its own
waits) and one wave in 64 leaves its cycles in a record of a global table (csrc/extract_kernels.hip, PhaseClk; no atomics).  The numbers are WALL cycles of a wave's life --
with 8 waves per SIMD sharing one vector issue port a phase's share of the life is its share of the SIMD's time, whatever the absolute figure.

    YGZF_LIBRARY=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_clk.so python tools/fast_phases.py                      # 752x480 synthetic + real-image clip + UHD
    YGZF_LIBRARY=... python tools/fast_phases.py fhd1920x1080_8lvl_4000feat

Per workload: the bench's own sub-batch on ONE context (isolated), both FAST threshold plans where the library would pick either.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "YGZF_LIBRARY" not in os.environ:
    os.environ["YGZF_LIBRARY"] = os.path.join(ROOT, "orb_ygz_slam_amd", "lib_ab", "libygzf_clk.so")
import bench  # noqa: E402
from orb_ygz_slam_amd import Extractor  # noqa: E402

FAST_PHASES = ["stage window (record load, LDS-DMA, wait)", "pass 1 (byte-sliced 9/16 test of every quad, quad list)", "expansion (quad list -> corner list)",
               "score-map zeroing", "corner score (fast9_arc_score)", "3x3 NMS + threshold choice + output"]
DESC_PHASES = ["stage window (record load, LDS-DMA, wait)", "intensity centroid (31x31 disc, wave sum)", "fastAtan2", "row blur (43 x 37, dot4)",
               "sincos of the angle (double)", "rotate pattern + column blur at 512 points + ballots", "descriptor / KeyPoint stores"]


def table(name, counters, phases):
    waves = int(counters[15])
    if not waves:
        return {"kernel": name, "waves_sampled": 0}
    cyc = [float(counters[k]) / waves for k in range(len(phases))]
    tot = sum(cyc)
    rows = [{"phase": p, "cycles_per_wave": round(c, 1), "share": round(c / tot, 4)} for p, c in zip(phases, cyc)]
    out = {"kernel": name, "waves_sampled": waves, "cycles_per_wave": round(tot, 1), "us_per_wave_at_2.4GHz": round(tot / 2400.0, 3), "phases": rows}
    if name == "k_fast_tab":
        out["pass1_runs_per_wave"] = round(float(counters[14]) / waves, 4)
    return out


def run(wl, frames, label, plan=None):
    w, h, nl, sf, nf, ini, mn = bench.WORKLOADS[wl][:7]
    B = len(frames)
    ex = Extractor(nf, sf, nl, ini, mn, max_width=w, max_height=h, max_batch=B)
    if plan:
        ex.set_fast_plan(plan)
    for _ in range(3):
        ex.extract_batch_host(frames)
    ex.sync()
    ex.phase_clocks(0, True); ex.phase_clocks(1, True)
    reps = 5
    ex.profile_enable(True); ex.profile_reset()
    for _ in range(reps):
        ex.extract_batch_host(frames)
    ex.sync()
    prof = {k: round(1e3 * ms / n, 1) for k, (ms, n) in ex.profile_read().items() if n}
    fast, desc = ex.phase_clocks(0, True), ex.phase_clocks(1, True)
    stats = ex.fast_stats()
    res = {"workload": wl, "clip": label, "frames_per_launch": B, "fast_plan": {1: "one pass at minTh", 2: "iniTh first"}.get(ex.fast_plan()),
           "corner_quads_per_pass1_run": round(stats[0], 2), "pass1_runs_per_cell": round(stats[1], 4),
           "keypoints_per_frame": round(float(ex.batch_counts().mean()), 1),
           "kernel_us_per_launch_instrumented": prof, "k_fast_tab": table("k_fast_tab", fast, FAST_PHASES), "k_describe": table("k_describe", desc, DESC_PHASES)}
    ex.close()
    return res


def show(r):
    print("== %s, %s, %d frames per launch, plan: %s; %.1f corner quads per pass-1 run, %.3f pass-1 runs per cell, %.0f keypoints per frame" %
          (r["workload"], r["clip"], r["frames_per_launch"], r["fast_plan"], r["corner_quads_per_pass1_run"], r["pass1_runs_per_cell"], r["keypoints_per_frame"]))
    print("   instrumented kernel times (us per launch):", r["kernel_us_per_launch_instrumented"])
    for k in ("k_fast_tab", "k_describe"):
        t = r[k]
        if not t.get("waves_sampled"):
            print("   %s: no waves sampled" % k)
            continue
        print("   %s: %d waves sampled, %.0f cycles = %.2f us per wave%s" % (k, t["waves_sampled"], t["cycles_per_wave"], t["us_per_wave_at_2.4GHz"],
                                                                           (", %.3f pass-1 runs per wave" % t["pass1_runs_per_wave"]) if "pass1_runs_per_wave" in t else ""))
        for row in t["phases"]:
            print("      %6.1f %%  %8.0f cycles  %s" % (100 * row["share"], row["cycles_per_wave"], row["phase"]))


def main():
    wls = sys.argv[1:] or ["euroc752x480_8lvl_1000feat", "euroc752x480_test1png", "uhd3840x2160_12lvl_8000feat"]
    out = []
    for name in wls:
        real = name == "euroc752x480_test1png"
        wl = "euroc752x480_8lvl_1000feat" if real else name
        w, h = bench.WORKLOADS[wl][:2]
        B = bench.SHAPES[wl][0]
        if real:
            base = bench.make_frames_test1png(96, w, h)
        else:
            base = bench.make_frames(min(B, 8 if "uhd" in wl else 24 if "fhd" in wl else B), w, h, seed0=1000)
        frames = np.ascontiguousarray(np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B])
        for plan in (None, 1, 2):
            r = run(wl, frames, "real-image clip (test1.png)" if real else "synthetic clip", plan)
            r["plan_forced"] = plan
            show(r)
            out.append(r)
    dst = os.path.join(ROOT, "gpurun_out", "fast_phases.json")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    json.dump(out, open(dst, "w"), indent=1)
    print("written:", dst)


if __name__ == "__main__":
    main()
