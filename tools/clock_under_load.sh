#!/bin/bash
# tools/clock_under_load.sh -- shader clock the chip holds under each kernel: GRBM_GUI_ACTIVE (GPU-busy cycles inside the dispatch window)
# divided by the dispatch duration, one stream at a time (run on the GPU box from the repo root; --pmc with --kernel-trace only).
REPO=$(pwd); OUT=$REPO/gpurun_out/clock; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk -o clk -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile --streams ${STREAMS:-1} > $OUT/run.log 2>&1
f=$(find /tmp/clk -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/clock.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ygzf::", "")
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    a = acc[name]
    a[0] += float(r["Counter_Value"]); a[1] += dur; a[2] += 1
print("kernel, launches, mean_us, GRBM_GUI_ACTIVE per launch, cycles per ns (= GHz if one counter instance)")
for k, (c, d, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%s, %d, %.1f, %.0f, %.3f" % (k, n, d / n / 1e3, c / n, c / d))
PY
