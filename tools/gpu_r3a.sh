bash tools/profile_round.sh r03_b > gpurun_out/prof_r03_b.log 2>&1
tail -5 gpurun_out/prof_r03_b.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_b_bench_default.json 2> gpurun_out/r03_b_bench_default.err
tail -c 600 gpurun_out/r03_b_bench_default.json
bash tools/round_numbers.sh > gpurun_out/r03_b_round_numbers.txt 2>&1
cat gpurun_out/r03_b_round_numbers.txt
