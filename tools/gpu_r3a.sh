timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_golden.py tests/test_gpu_blur_modes.py tests/test_gpu_soak.py tests/test_gpu_dso.py tests/test_gpu_grid_detectors.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -3
run() { python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('kernels_isolated_avg_us') or {}).get('k_describe'))"; }
for i in 1 2 3; do
YGZF_LIBRARY=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_HEAD.so run old
run new
done
