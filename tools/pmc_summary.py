#!/usr/bin/env python3
"""tools/pmc_summary.py <prof dir> <tag> -- condense the rocprofv3 outputs of tools/profile_round.sh into
  <tag>_kernel_stats.csv / <tag>_all_kernel_stats.csv   rocprofv3 --stats of the bench command / of tools/all_kernels.py
  <tag>_pmc_hbm.csv / <tag>_all_pmc_hbm.csv             FETCH_SIZE + WRITE_SIZE per kernel launch
  <tag>_pmc_sq.csv                                      SQ occupancy / wait counters of the bench kernels
  <tag>_pmc_insts.csv / <tag>_all_pmc_insts.csv         instruction counts per launch (VALU / SALU / LDS / VMEM, waves)
  traffic.json                                          HBM bytes per launch per kernel (read by bench.py: roofline.traffic)
  valu_mix.json                                         VALU instructions per frame + mean issue cost per instruction (bench.py: roofline_valu);
                                                        the issue cost comes from tools/valu_mix.py's classification of the kernel's ISA
Units: FETCH_SIZE / WRITE_SIZE are KB.  hbm_bytes = FETCH * 1024 * fetch_factor + WRITE * 1024 * write_factor with the factors of
profiles/hbm_calibration.json (tools/hbm_calib.sh: known 1 GiB streams in the access widths of these kernels -- dword, dwordx4 and
LDS-DMA reads all report exactly HALF their bytes on gfx950, writes report their bytes: fetch_factor 2.0, write_factor 1.0).  The raw
counter bytes stay in the CSV beside the corrected ones."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    n = m.group(1) if m else name.split("(")[0]
    return "k_pyr_resize" if n.startswith("k_pyr_resize") else n


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r.get("Dispatch_Id"), r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r.get("Dispatch_Id")] = short(r["Kernel_Name"])
        for (disp, cn), v in per_dispatch.items():
            acc[names[disp]][cn].append(v)
    return acc


def mean(v):
    return sum(v) / len(v) if v else 0.0


def calibration():
    """(fetch factor, write factor) of profiles/hbm_calibration.json: every read width these kernels use calibrated to the same factor."""
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "hbm_calibration.json")
    if not os.path.exists(f):
        return 1.0, 1.0, "uncalibrated (profiles/hbm_calibration.json missing)"
    c = json.load(open(f))
    ff = [c["fetch_factor"][k] for k in ("dword", "dwordx4", "ldsdma16") if k in c["fetch_factor"]]
    wf = [c["write_factor"][k] for k in ("dword", "dwordx4") if k in c["write_factor"]]
    return round(sum(ff) / len(ff), 3), round(sum(wf) / len(wf), 3), "profiles/hbm_calibration.json"


def hbm_table(out, tag, prefix, what):
    fetch, write = counters(os.path.join(out, prefix + "fetch")), counters(os.path.join(out, prefix + "write"))
    ff, wf, src = calibration()
    traffic = {}
    with open(os.path.join(out, "%s_%spmc_hbm.csv" % (tag, prefix)), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), %s\n" % what)
        f.write("# units: KB per launch (mean over launches); raw_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024; hbm_bytes = FETCH_SIZE * 1024 * %.3f + WRITE_SIZE * 1024 * %.3f (%s)\n" % (ff, wf, src))
        f.write("kernel,launches,FETCH_SIZE_KB,WRITE_SIZE_KB,raw_bytes_per_launch,hbm_bytes_per_launch\n")
        for k in sorted(set(fetch) | set(write)):
            fv, wv = fetch.get(k, {}).get("FETCH_SIZE", []), write.get(k, {}).get("WRITE_SIZE", [])
            traffic[k] = int(mean(fv) * 1024 * ff + mean(wv) * 1024 * wf)
            f.write("%s,%d,%.1f,%.1f,%d,%d\n" % (k, max(len(fv), len(wv)), mean(fv), mean(wv), int((mean(fv) + mean(wv)) * 1024), traffic[k]))
    return traffic


def insts_table(out, tag, prefix, what):
    ins = counters(os.path.join(out, prefix + "insts"))
    cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"]
    with open(os.path.join(out, "%s_%spmc_insts.csv" % (tag, prefix)), "w") as f:
        f.write("# rocprofv3 --pmc " + " ".join(cols) + " (one pass, --kernel-trace only), %s; mean per launch\n" % what)
        f.write("kernel,launches," + ",".join(cols) + ",valu_per_wave\n")
        for k in sorted(ins):
            n = max(len(v) for v in ins[k].values())
            vals = [mean(ins[k].get(c, [])) for c in cols]
            f.write(k + "," + str(n) + "," + ",".join("%.0f" % v for v in vals) + ",%.1f\n" % (vals[1] / max(vals[0], 1)))
    return ins


def copy_stats(out, tag, sub, name):
    stats = glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(out, name), "w", newline="") as f:
            csv.writer(f, quoting=csv.QUOTE_ALL).writerows(rows)


def main():
    out, tag = sys.argv[1], sys.argv[2]
    sub = int(os.environ.get("YGZF_PROFILE_SUB_BATCH", "256"))
    copy_stats(out, tag, "stats", tag + "_kernel_stats.csv")
    copy_stats(out, tag, "all_stats", tag + "_all_kernel_stats.csv")
    traffic = hbm_table(out, tag, "", "python bench.py --steps 3 --warmup 1 (sub-batches of 256 frames)")
    hbm_table(out, tag, "all_", "python tools/all_kernels.py 2")
    sq = counters(os.path.join(out, "sq"))
    with open(os.path.join(out, tag + "_pmc_sq.csv"), "w") as f:
        cols = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"]
        f.write("# rocprofv3 --pmc " + " ".join(cols) + " (one pass, --kernel-trace only); mean per launch.  SQ_WAVE_CYCLES / SQ_WAIT_* count in units of 4 cycles\n")
        f.write("# (calibrated with tools/micro/valu_peak.hip: a wave issuing 128 k instructions at 5.7 cycles each reports 166 k); SQ_ACTIVE_INST_VALU counts instructions\n")
        f.write("kernel,launches," + ",".join(cols) + "\n")
        for k in sorted(sq):
            n = max(len(v) for v in sq[k].values())
            f.write(k + "," + str(n) + "," + ",".join("%.0f" % mean(sq[k].get(c, [])) for c in cols) + "\n")
    ins = insts_table(out, tag, "", "python bench.py --steps 3 --warmup 1")
    insts_table(out, tag, "all_", "python tools/all_kernels.py 2")
    ff, wf, src = calibration()
    json.dump({"_source": "profiles/%s_pmc_hbm.csv (rocprofv3 PMC, HBM bytes per launch of one 256-frame sub-batch), CORRECTED: FETCH_SIZE x 1024 x %.3f + "
                          "WRITE_SIZE x 1024 x %.3f (%s)" % (tag, ff, wf, src), "fetch_factor": ff, "write_factor": wf,
               "euroc752x480_8lvl_1000feat": traffic}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    # VALU mix: instructions per frame from the counters, mean issue cost per instruction from the ISA classification (tools/valu_mix.py)
    cost = {}
    cf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "valu_issue_cost.json")
    if os.path.exists(cf):
        cost = json.load(open(cf))
    mix = {}
    for k in ins:
        v = mean(ins[k].get("SQ_INSTS_VALU", []))
        if v > 0 and k in cost:
            per_frame = v / sub * (7 if k == "k_pyr_resize" else 1)      # 7 resize launches per frame
            mix[k] = {"valu_insts_per_frame": round(per_frame, 1), "cycles_per_inst": cost[k]["cycles_per_inst"],
                      "valu_insts_per_launch": round(v, 1), "waves_per_launch": round(mean(ins[k].get("SQ_WAVES", [])), 1)}
    json.dump({"_source": "SQ_INSTS_VALU of profiles/%s_pmc_insts.csv / %d frames per launch; cycles_per_inst from profiles/valu_issue_cost.json "
                          "(tools/valu_mix.py: ISA of the kernel classified with the issue rates of profiles/micro/r02_valu_issue_rates.txt)" % (tag, sub),
               "euroc752x480_8lvl_1000feat": mix}, open(os.path.join(out, "valu_mix.json"), "w"), indent=1)
    print(open(os.path.join(out, tag + "_pmc_hbm.csv")).read())
    print(open(os.path.join(out, tag + "_pmc_insts.csv")).read())


if __name__ == "__main__":
    main()
