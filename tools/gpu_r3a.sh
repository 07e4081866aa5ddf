run() { python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('kernels_isolated_avg_us') or {}).get('k_pyr_resize'))"; }
run p4
for k in 1 2; do YGZF_LIBRARY=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_p$k.so run p$k; done
run p4
for k in 1 2; do YGZF_LIBRARY=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_p$k.so run p$k; done
YGZF_LIBRARY=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_p2.so timeout 600 python -m pytest tests/test_gpu_extract.py -x -q -p no:cacheprovider 2>&1 | tail -1
