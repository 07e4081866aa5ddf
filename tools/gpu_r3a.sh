mkdir -p gpurun_out/r3a
timeout 600 python -m pytest tests/test_gpu_fast_kernels.py tests/test_gpu_fast_plans.py tests/test_gpu_extract.py -x -q 2>&1 | tail -5
timeout 300 python tools/fast_ab.py 256 10 > gpurun_out/r3a/fast_ab_tab.json 2> gpurun_out/r3a/fast_ab.err; cat gpurun_out/r3a/fast_ab_tab.json; tail -3 gpurun_out/r3a/fast_ab.err
