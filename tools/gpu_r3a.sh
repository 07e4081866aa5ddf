timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -3
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernels_isolated_avg_us'))"
