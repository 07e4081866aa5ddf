timeout 600 python -m pytest tests/test_gpu_mgpu.py -x -q 2>&1 | tail -2
timeout 300 python tools/mgpu_rate.py --literal 2>&1 | tail -2
