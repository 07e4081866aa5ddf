timeout 600 python -m pytest tests/test_gpu_boundary.py -x -q -p no:cacheprovider 2>&1 | tail -25
