mkdir -p gpurun_out/r3a
timeout 120 tools/micro/bin/valu_peak f64 "s_add" salu > gpurun_out/r3a/valu_rates_f64_salu.txt 2>&1; cat gpurun_out/r3a/valu_rates_f64_salu.txt
