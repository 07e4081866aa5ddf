#!/usr/bin/env python3
"""tools/hbm_calib_summary.py <dir> -- FETCH_SIZE / WRITE_SIZE of tools/micro/hbm_calib's streams against their known byte counts ->
<dir>/hbm_calibration.json: {"fetch_factor": {pattern: bytes / (FETCH_SIZE * 1024)}, "write_factor": {...}, raw numbers}."""
import collections
import csv
import glob
import json
import os
import re
import sys

N = 1 << 30
ROWS = (N // 752 // 21) * 21
KNOWN = {"calib_read_dword": N, "calib_read_dwordx4": N, "calib_read_ldsdma16": N, "calib_read_ldsdma16_rows48": ROWS * 48,
         "calib_write_dword": N, "calib_write_dwordx4": N, "calib_write_byte": N // 4}


def counters(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = re.search(r"(calib_[a-z0-9_]+)", r["Kernel_Name"]).group(1) if "calib_" in r["Kernel_Name"] else None
        for disp, v in per.items():
            if names[disp]:
                acc[names[disp]].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    out = sys.argv[1]
    fetch, write = counters(os.path.join(out, "fetch"), "FETCH_SIZE"), counters(os.path.join(out, "write"), "WRITE_SIZE")
    res = {"_doc": "rocprofv3 FETCH_SIZE / WRITE_SIZE (KB) of known-size streams (tools/micro/hbm_calib.hip, 1 GiB each, separate --pmc passes); "
                   "factor = known bytes / (counter * 1024): multiply a kernel's raw counter bytes by the factor of its dominant access width. "
                   "read_ldsdma16_rows48 requests 48-byte rows at byte-unaligned starts 752 bytes apart (the FAST window pattern): its factor relates "
                   "the counter to the bytes REQUESTED, the cache lines touched are more",
           "known_bytes": KNOWN, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "fetch_factor": {}, "write_factor": {}}
    for k, b in KNOWN.items():
        if k.startswith("calib_read") and fetch.get(k):
            res["fetch_factor"][k[len("calib_read_"):]] = round(b / (fetch[k] * 1024.0), 4)
        if k.startswith("calib_write") and write.get(k):
            res["write_factor"][k[len("calib_write_"):]] = round(b / (write[k] * 1024.0), 4)
    json.dump(res, open(os.path.join(out, "hbm_calibration.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
