#!/bin/bash
# round 5, GPU call 2: register-resident aligner -- parity tests, rates, latency; hand-over stress; upload / partition tests
O=gpurun_out/r05c2
mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_align.py tests/test_align_ref_golden.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests_align.txt 2>&1
tail -3 $O/tests_align.txt
timeout 600 python -m pytest tests/test_gpu_handover.py tests/test_gpu_partition_upload.py -x -q -m gpu > $O/tests_new.txt 2>&1
tail -3 $O/tests_new.txt
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in (d.get('kernels') or {}).items()})"; }
python bench.py --no-cpu-baseline --no-extras --align --steps 5 2>&1 | p align_default | tee -a $O/align.txt
YGZF_SIA_LDS_CAP=0 python bench.py --no-cpu-baseline --no-extras --align --steps 5 2>&1 | p align_nocap | tee -a $O/align.txt
YGZF_SIA_LDS_CAP=100 python bench.py --no-cpu-baseline --no-extras --align --steps 5 2>&1 | p align_cap100 | tee -a $O/align.txt
python tools/sia_phases.py 2>&1 | tail -2 | tee -a $O/align.txt
YGZF_SIA_DEBUG=1 python tools/sia_phases.py 2>&1 | tail -4 | tee -a $O/align.txt
for b in 1 16; do python bench.py --no-cpu-baseline --no-extras --align --streams 1 --sub-batch $b --batch $b --steps 50 --warmup 5 2>&1 | p "align_lat_b$b" | tee -a $O/align.txt; done
