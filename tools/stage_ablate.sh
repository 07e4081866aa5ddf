#!/bin/bash
# usage: tools/stage_ablate.sh ENVVAR kernel "stages..."   (debug: per-stage early exit timing of one kernel)
for st in $3; do
  export $1=$st
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('stage', os.environ['$1'], d['kernels']['$2']['avg_us'])"
done
