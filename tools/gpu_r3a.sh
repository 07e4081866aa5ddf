timeout 900 python -m pytest tests/test_gpu_match.py tests/test_gpu_frustum.py tests/test_gpu_boundary.py tests/test_gpu_shells.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -4
YGZF_FUZZ_SEEDS=100 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -p no:cacheprovider -k "projected or frustum or search_by" 2>&1 | grep -v "^$" | tail -3
YGZF_MATCH_DEBUG=1 timeout 200 python tools/call_latency.py 2>&1 | grep "mode 1\|cur, last" | tail -3
timeout 200 python tools/call_latency.py 2>/dev/null | grep -i "search\|frustum"
