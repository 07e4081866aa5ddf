timeout 600 python -m pytest tests/test_gpu_shells.py tests/test_gpu_boundary.py tests/test_gpu_align.py -x -q 2>&1 | tail -3
cat gpurun_out/shell_latency.txt
