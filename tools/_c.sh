timeout 600 python -m pytest tests/test_gpu_mgpu.py tests/test_gpu_stereo.py -x -q 2>&1 | tail -15
