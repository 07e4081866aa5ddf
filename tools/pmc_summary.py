#!/usr/bin/env python3
"""tools/pmc_summary.py <prof dir> <tag> -- condense the rocprofv3 outputs of tools/profile_round.sh into
<prof dir>/<tag>_kernel_stats.csv, <tag>_pmc_hbm.csv, <tag>_pmc_sq.csv and traffic.json (bytes per launch per kernel).
Units: FETCH_SIZE / WRITE_SIZE are KB; hbm_bytes = (FETCH + WRITE) * 1024 (no gfx950 1/2 correction: these kernels read bytes /
dwords, not 16 B per lane streams -- calibrated on k_pyr_resize's known bytes, see profiles/r01_a_pmc_hbm_b256.csv)."""
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


def main():
    out, tag = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as f:
            csv.writer(f, quoting=csv.QUOTE_ALL).writerows(rows)
    fetch, write, sq = counters(os.path.join(out, "fetch")), counters(os.path.join(out, "write")), counters(os.path.join(out, "sq"))
    traffic = {}
    with open(os.path.join(out, tag + "_pmc_hbm.csv"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), python bench.py --steps 5 --warmup 2\n")
        f.write("# units: KB per launch (mean over launches); hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024\n")
        f.write("kernel,launches,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_per_launch\n")
        for k in sorted(set(fetch) | set(write)):
            fv, wv = fetch.get(k, {}).get("FETCH_SIZE", []), write.get(k, {}).get("WRITE_SIZE", [])
            fm = sum(fv) / len(fv) if fv else 0.0
            wm = sum(wv) / len(wv) if wv else 0.0
            traffic[k] = int((fm + wm) * 1024)
            f.write("%s,%d,%.1f,%.1f,%d\n" % (k, max(len(fv), len(wv)), fm, wm, traffic[k]))
    with open(os.path.join(out, tag + "_pmc_sq.csv"), "w") as f:
        cols = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"]
        f.write("# rocprofv3 --pmc " + " ".join(cols) + " (one pass, --kernel-trace only); mean per launch\n")
        f.write("kernel,launches," + ",".join(cols) + "\n")
        for k in sorted(sq):
            n = max(len(v) for v in sq[k].values())
            f.write(k + "," + str(n) + "," + ",".join("%.0f" % (sum(sq[k].get(c, [0])) / max(len(sq[k].get(c, [0])), 1)) for c in cols) + "\n")
    json.dump({"_source": "profiles/%s_pmc_hbm.csv (rocprofv3 PMC, HBM bytes per launch at the default bench configuration)" % tag,
               "euroc752x480_8lvl_1000feat": traffic}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(open(os.path.join(out, tag + "_pmc_hbm.csv")).read())


if __name__ == "__main__":
    main()
