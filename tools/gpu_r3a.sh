timeout 600 python -m pytest tests/test_gpu_extract.py -x -q -p no:cacheprovider -k "real_image or bench_clip" 2>&1 | tail -8
