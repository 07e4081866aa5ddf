p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
python bench.py --no-cpu-baseline --no-profile --no-extras --streams 1 --batch 256 2>&1 | p "s1_b256"
python bench.py --no-cpu-baseline --no-profile --no-extras --workload fhd1920x1080_8lvl_4000feat --steps 8 2>&1 | p fhd
python bench.py --no-cpu-baseline --no-profile --no-extras --workload uhd3840x2160_12lvl_8000feat --steps 6 2>&1 | p uhd
python bench.py --no-cpu-baseline --no-profile --no-extras --workload uhd3840x2160_12lvl_8000feat --steps 6 --stereo 2>&1 | p uhd_stereo
for b in 1 4 16 64; do python bench.py --no-cpu-baseline --no-profile --no-extras --streams 1 --sub-batch $b --batch $b --steps 50 --warmup 5 2>&1 | p "lat_b$b"; done
python bench.py --no-cpu-baseline --no-extras --streams 1 --sub-batch 1 --batch 1 --steps 50 --warmup 5 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lat kernels', {k:v['avg_us'] for k,v in d['kernels'].items()})"
