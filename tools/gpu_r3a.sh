mkdir -p gpurun_out/r3a
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench2.json 2> gpurun_out/r3a/bench2.err; tail -3 gpurun_out/r3a/bench2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3a/bench2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['timed_region_s'], d.get('value_end_to_end'), d['end_to_end'])
print(d['kernels_isolated_avg_us'])
print(d['roofline']); print(d['roofline_valu'])
for k,v in d['other_workloads'].items(): print(k, v['value'], v['roofline'], v.get('fast_plan'), v['keypoints_per_frame'], v['matches_per_frame'])
print(d['cpu_baseline']); print(d.get('libfast_sse2_anchor'))
PY
