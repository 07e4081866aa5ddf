#!/bin/bash
O=gpurun_out/r05c5
mkdir -p $O
python -c "import torch" 2>/dev/null
for o in 1 0 1 0; do YGZF_MGPU_ORDER=$o timeout 200 python tools/mgpu_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/mgpu_ab.jsonl; done
