#!/bin/bash
# round 5, GPU call 3: the whole GPU tier, the driver's bench line, shell / boundary latencies
O=gpurun_out/r05c3
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests_gpu.txt 2>&1
tail -4 $O/tests_gpu.txt
cp gpurun_out/shell_latency.txt $O/ 2>/dev/null; cp gpurun_out/boundary_latency.txt $O/ 2>/dev/null
cat $O/shell_latency.txt $O/boundary_latency.txt
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "e2e", d["value_end_to_end"], d["roofline_pcie"], "mgpu", d["mgpu_end_to_end"]["value"], d["mgpu_end_to_end"]["value_pageable"], "fallbacks", d.get("match_serial_fallback_pairs"))
for k,v in d["other_workloads"].items(): print(k, v["value"], v.get("value_end_to_end"), v["kernels"])
for k,v in d["mgpu_literal_configs"].items(): print(k, v["ms_per_call"], v["one_unit"])
print(d["config"])
PY
