timeout 900 python -m pytest tests/test_gpu_shells.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -4
