#!/bin/bash
# A/B: k_octree workgroups of 1024 / 512 / 256 threads x LDS allotments, correctness first (octree parity tests on each build), then the default bench
O=gpurun_out/r05oct
mkdir -p $O
python -c "import torch" 2>/dev/null
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()}, d['kernels_isolated_avg_us'].get('k_octree'))"; }
for blk in 512 256; do
  L=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_oct$blk.so
  echo "== tests block $blk"; YGZF_LIBRARY=$L timeout 300 python -m pytest tests/test_gpu_octree_plans.py tests/test_gpu_extract.py -x -q -m gpu 2>&1 | tail -2
done | tee $O/tests.txt
python bench.py --no-cpu-baseline --no-extras --steps 6 2>&1 | p "blk1024_lds71" | tee -a $O/oct.txt
for blk in 512 256; do
  L=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_oct$blk.so
  for kb in 71 56 44 32; do
    YGZF_LIBRARY=$L YGZF_OCT_LDS_KB=$kb python bench.py --no-cpu-baseline --no-extras --steps 6 2>&1 | p "blk${blk}_lds$kb" | tee -a $O/oct.txt
  done
done
