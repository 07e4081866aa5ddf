python bench.py --steps 20 --warmup 5 > gpurun_out/r03_c_bench_default.json 2> gpurun_out/r03_c_bench_default.err
tail -c 400 gpurun_out/r03_c_bench_default.json
