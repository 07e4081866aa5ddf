timeout 900 python -m pytest tests/test_gpu_errors.py tests/test_gpu_shells.py tests/test_gpu_boundary.py tests/test_gpu_bow.py -x -q -p no:cacheprovider 2>&1 | tail -15
