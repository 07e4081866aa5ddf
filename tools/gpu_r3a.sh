timeout 300 python -m pytest tests/test_gpu_match.py tests/test_gpu_shells.py -x -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -14
timeout 200 python tools/call_latency.py 2>/dev/null | head -8
