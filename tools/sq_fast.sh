#!/bin/bash
# tools/sq_fast.sh -- SQ counters of the FAST and describe kernels in two passes (run on the GPU box from the repo root; --pmc with --kernel-trace only).
# Output: gpurun_out/sq_fast/cc_{a,b}.csv (the k_fast_* / k_describe rows of rocprofv3's counter_collection.csv) and summary.txt (mean per launch);
# profiles/micro/ keeps a copy.   PASSES=b runs only the LDS pass.
REPO=$(pwd); OUT=$REPO/gpurun_out/sq_fast; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python $REPO/bench.py --steps 2 --warmup 1 --passes 1 --no-cpu-baseline --no-extras --no-profile --streams 1"
PASSES=${PASSES:-a b}
for p in $PASSES; do
  if [ $p = a ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
  else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM"; fi
  rm -rf /tmp/sq$p
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/sq$p -o $p -- $B > $OUT/$p.log 2>&1
  f=$(find /tmp/sq$p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -i "k_fast\|k_describe\|Counter_Name" $f > $OUT/cc_$p.csv
done
python - $OUT <<'PY' | tee $OUT/summary.txt
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(sys.argv[1] + "/cc_*.csv")):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].split("::")[-1], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kern, ctr), (s, n) in sorted(acc.items()):
    print("%-28s %-24s mean per launch %14.0f  (%d launches)" % (kern, ctr, s / n, n))
PY
