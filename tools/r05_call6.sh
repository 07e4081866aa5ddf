#!/bin/bash
O=gpurun_out/r05c6
mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_fast_plans.py tests/test_gpu_fast_kernels.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests.txt
timeout 300 python tools/fast_pretest_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/pretest_ab.jsonl
