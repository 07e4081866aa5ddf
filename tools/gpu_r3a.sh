python bench.py --steps 20 --warmup 5 > gpurun_out/r03_c_bench_default.json 2> gpurun_out/r03_c_bench_default.err
tail -c 300 gpurun_out/r03_c_bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --passes 1 --no-cpu-baseline --no-extras > /tmp/st.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r03_c; cp $(find /tmp/st -name "*kernel_stats.csv") gpurun_out/prof_r03_c/r03_c_kernel_stats_raw.csv
head -12 gpurun_out/prof_r03_c/r03_c_kernel_stats_raw.csv | cut -c1-150
