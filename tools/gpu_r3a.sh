mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_e_bench_default.json 2> gpurun_out/r03_e_bench_default.err
tail -c 300 gpurun_out/r03_e_bench_default.json
bash tools/round_numbers.sh > gpurun_out/r03_e_round_numbers.txt 2>&1
cat gpurun_out/r03_e_round_numbers.txt
timeout 600 python -m pytest tests/test_gpu_shells.py -x -q -p no:cacheprovider -k latency -s 2>&1 | grep -E "^[a-z_]+ [0-9.]+ [0-9.]+|passed|failed" | tee gpurun_out/r03_e_shell_latency.txt
bash tools/profile_round.sh r03_e > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log
