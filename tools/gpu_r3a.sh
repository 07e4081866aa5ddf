run() { python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items() if 'match' in k}, {k:v for k,v in (d.get('kernels_isolated_avg_us') or {}).items() if 'match' in k})"; }
run base
YGZF_MATCH_PLAN=1 run plan1
YGZF_MATCH_PLAN=2 run plan2
YGZF_MATCH_PLAN=3 run plan3
