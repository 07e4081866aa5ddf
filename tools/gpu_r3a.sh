run() { python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('kernels_isolated_avg_us') or {}).get('k_describe'))"; }
run full
YGZF_LIBRARY=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_skipborder.so run skipborder
run full
YGZF_LIBRARY=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_skipborder.so run skipborder
