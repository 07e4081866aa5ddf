timeout 900 python -m pytest tests/test_gpu_extract.py -x -q -p no:cacheprovider 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2 --passes 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], (d.get('kernels_isolated_avg_us') or {}).get('k_pyr_resize'))"; }
YGZF_PYR_TAIL_FROM=0 run notail
run tail3
YGZF_PYR_TAIL_FROM=4 run tail4
YGZF_PYR_TAIL_FROM=2 run tail2
YGZF_PYR_STRIPS=24 run tail3s24
YGZF_PYR_TAIL_FROM=0 run notail
run tail3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --no-cpu-baseline --no-extras --no-profile --steps 2 --warmup 1 --passes 1 --streams 1 > /dev/null 2>&1
f=$(find gpurun_out/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'pyr' in n:
        key=(n.split('(')[0][-28:], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
        agg[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    v.sort(); print(k, len(v), 'median us', v[len(v)//2]/1e3, 'min', v[0]/1e3)
PY
rm -rf gpurun_out/kt
