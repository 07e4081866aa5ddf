#!/bin/bash
# tools/r04_check.sh <tag> [what...] -- one GPU call of round 4: the GPU test suite, then short bench lines of the workloads named
#   what: tests default uhd fhd align oct   (default: all)
TAG=${1:-x}; shift
WHAT=${@:-tests default uhd fhd align oct}
OUT=gpurun_out/r04/$TAG
mkdir -p $OUT
for w in $WHAT; do
  case $w in
    tests)   timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log;;
    default) timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/default.json 2> $OUT/default.err;;
    uhd)     timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload uhd3840x2160_12lvl_8000feat --stereo --distinct 8 > $OUT/uhd.json 2> $OUT/uhd.err;;
    fhd)     timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload fhd1920x1080_8lvl_4000feat --distinct 24 > $OUT/fhd.json 2> $OUT/fhd.err;;
    align)   timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --align > $OUT/align.json 2> $OUT/align.err;;
    oct)     timeout 200 python tools/oct_phases.py uhd3840x2160_12lvl_8000feat 64 2>&1 | tail -12 > $OUT/oct_uhd.txt
             timeout 200 python tools/oct_phases.py fhd1920x1080_8lvl_4000feat 128 2>&1 | tail -8 > $OUT/oct_fhd.txt;;
  esac
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d.get("ms_per_step"), "iso", d.get("kernels_isolated_avg_us"), "pipe", d["roofline"].get("pipeline_frac"))
        print("   in-bench", {k: v["avg_us"] for k, v in d.get("kernels", {}).items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
