mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r3a/pytest.log 2>&1; tail -4 gpurun_out/r3a/pytest.log
