run() { python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('kernels_isolated_avg_us') or {}).get('k_match_last'))"; }
run base
YGZF_MATCH_SPLIT=2 run split2
YGZF_MATCH_SPLIT=4 run split4
run base
YGZF_MATCH_SPLIT=2 run split2
