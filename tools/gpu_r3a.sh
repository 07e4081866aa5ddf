timeout 900 python -m pytest tests/test_gpu_align.py tests/test_gpu_repeat.py tests/test_gpu_soak.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -6
python bench.py --no-cpu-baseline --no-extras --align --steps 6 --warmup 2 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
