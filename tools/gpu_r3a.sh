timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider -k "sparse_img_align" 2>&1 | tail -12; cat gpurun_out/align_fuzz_report.json
