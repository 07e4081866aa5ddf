timeout 900 python -m pytest tests/test_gpu_repeat.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -4
YGZF_REPEATS=300 timeout 900 python -m pytest tests/test_gpu_repeat.py -x -q -p no:cacheprovider -k two_contexts 2>&1 | grep -v "^$" | tail -3
