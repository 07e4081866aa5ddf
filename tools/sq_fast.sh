#!/bin/bash
# tools/sq_fast.sh -- SQ counters of the FAST kernel in two passes (run on the GPU box from the repo root; --pmc with --kernel-trace only).
# Output: gpurun_out/sq_fast/cc_{a,b}.csv (the k_fast_quads rows of rocprofv3's counter_collection.csv); profiles/micro/ keeps a copy.
REPO=$(pwd); OUT=$REPO/gpurun_out/sq_fast; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --streams 1"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/sqa -o a -- $B > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/sqb -o b -- $B > $OUT/b.log 2>&1
for p in a b; do f=$(find /tmp/sq$p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -i "fast\|Counter_Name" $f > $OUT/cc_$p.csv; done
