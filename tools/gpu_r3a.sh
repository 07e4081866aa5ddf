timeout 1500 python -m pytest tests/test_gpu_extract.py tests/test_gpu_golden.py tests/test_gpu_blur_modes.py tests/test_gpu_soak.py tests/test_gpu_dso.py tests/test_gpu_grid_detectors.py tests/test_gpu_shells.py tests/test_gpu_repeat.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -3
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels_isolated_avg_us'))"
python bench.py --no-cpu-baseline --no-profile --no-extras --streams 1 --batch 1 --sub-batch 1 --steps 200 --warmup 10 2>&1 | tail -1 | cut -c1-140
