run() { python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()}, d.get('kernels_isolated_avg_us'))"; }
run base
YGZF_MATCH_DESC_GLOBAL=1 run desc_global
YGZF_OCT_LDS_KB=40 run oct40
YGZF_OCT_LDS_KB=30 run oct30
YGZF_OCT_LDS_KB=40 YGZF_MATCH_DESC_GLOBAL=1 run both
