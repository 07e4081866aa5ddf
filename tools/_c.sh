p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernels_isolated_avg_us'])"; }
for i in 1 2; do
YGZF_LIBRARY=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_ce0bc63.so python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3 2>&1 | p r03
python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3 2>&1 | p now
done
