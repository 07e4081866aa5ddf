YGZF_FUZZ_SEEDS=200 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -6
