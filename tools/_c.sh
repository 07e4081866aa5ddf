timeout 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_bow.py tests/test_gpu_shells.py -x -q 2>&1 | tail -30
cat gpurun_out/r04/boundary_latency.txt
