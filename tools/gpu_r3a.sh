YGZF_FUZZ_SEEDS=60 timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_fast_plans.py tests/test_gpu_fuzz.py -x -q -p no:cacheprovider 2>&1 | tail -2
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
lat() { for b in $2; do python bench.py --no-cpu-baseline --no-profile --no-extras --streams 1 --sub-batch $b --batch $b --steps 100 --warmup 10 2>&1 | p "$1 lat_b$b"; done; }
lat new "1 1"
YGZF_OCT_DEBUG=1 python bench.py --no-cpu-baseline --no-profile --no-extras --streams 1 --sub-batch 1 --batch 1 --steps 3 --warmup 1 2>&1 | grep "octree lvl" | tail -8
python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d.get('kernels_isolated_avg_us'))"
