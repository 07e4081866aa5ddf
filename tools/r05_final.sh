#!/bin/bash
# tools/r04_final.sh <tag> -- the measurement set of a round, one GPU call: rocprofv3 stats + PMC of the default bench and of every other kernel
# (profile_round.sh), of the FHD / UHD-stereo / align workloads (profile_shapes.sh), the driver's bench command, the one-frame numbers and the shells'
# latencies.  Everything lands in gpurun_out/final_<tag>/ ready to be copied into profiles/.
TAG=${1:-r05_e}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
bash tools/profile_shapes.sh $TAG uhd fhd align > $OUT/profile_shapes.log 2>&1
python tools/merge_traffic.py gpurun_out/prof_$TAG gpurun_out/shapes_$TAG $OUT/traffic.json >> $OUT/profile_shapes.log 2>&1
cp gpurun_out/prof_$TAG/${TAG}_*.csv gpurun_out/prof_$TAG/valu_mix.json $OUT/ 2>/dev/null
cp gpurun_out/shapes_$TAG/${TAG}_*.csv $OUT/ 2>/dev/null
cp $OUT/traffic.json profiles/traffic.json          # the bench line below reads it
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default.json 2> $OUT/bench.err
timeout 600 bash tools/round_numbers.sh > $OUT/${TAG}_round_numbers.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_shells.py tests/test_gpu_boundary.py -x -q -k "latency or reference_frame" > $OUT/shells.log 2>&1
cp gpurun_out/shell_latency.txt $OUT/${TAG}_shell_latency.txt
cp gpurun_out/boundary_latency.txt $OUT/${TAG}_boundary_latency.txt
tail -3 $OUT/shells.log
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "e2e", d["value_end_to_end"], "pipeline_frac", d["roofline"]["pipeline_frac"], "mgpu", d["mgpu_end_to_end"]["value"], d["mgpu_end_to_end"]["value_pageable"])
for k,v in d["other_workloads"].items(): print(k, v["value"], v.get("value_end_to_end"), v["roofline"]["pipeline_frac"], v["roofline"]["traffic"], v["roofline"]["isolated_avg_us"])
print(json.dumps(d["mgpu_literal_configs"])[:700])
PY
cat $OUT/${TAG}_round_numbers.txt
