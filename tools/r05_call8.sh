#!/bin/bash
O=gpurun_out/r05c8
mkdir -p $O
python -c "import torch" 2>/dev/null
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()}, d['kernels_isolated_avg_us'])"; }
for kb in 71 48 36 24; do YGZF_OCT_LDS_KB=$kb python bench.py --no-cpu-baseline --no-extras --steps 6 2>&1 | p "oct_lds_$kb" | tee -a $O/oct.txt; done
YGZF_OCT_PLAN=hist python bench.py --no-cpu-baseline --no-extras --steps 6 2>&1 | p "oct_hist" | tee -a $O/oct.txt
for w in 1 4; do YGZF_LIBRARY= python bench.py --no-cpu-baseline --no-extras --steps 6 --streams $((w+1)) 2>&1 | p "streams_$((w+1))" | tee -a $O/oct.txt; done
