timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_align.py tests/test_gpu_stereo.py tests/test_gpu_repeat.py -x -q -p no:cacheprovider 2>&1 | tail -3
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
lat() { for b in $2; do python bench.py --no-cpu-baseline --no-profile --no-extras --streams 1 --sub-batch $b --batch $b --steps 100 --warmup 10 2>&1 | p "$1 lat_b$b"; done; }
YGZF_PYR_STRIP_FRAMES=0 lat old "1 16"
lat s32 "1 2 8 16"
YGZF_PYR_STRIPS=48 lat s48 "1 16"
python bench.py --no-cpu-baseline --no-extras --streams 1 --sub-batch 1 --batch 1 --steps 50 --warmup 5 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lat kernels', {k:v['avg_us'] for k,v in d['kernels'].items()})"
