#!/bin/bash
# round 4, GPU call 1: measurement only (shapes beside the default workload)
mkdir -p gpurun_out/r04c1
timeout 200 python tools/oct_phases.py uhd3840x2160_12lvl_8000feat 64 > gpurun_out/r04c1/oct_uhd.txt 2>&1
timeout 200 python tools/oct_phases.py fhd1920x1080_8lvl_4000feat 128 > gpurun_out/r04c1/oct_fhd.txt 2>&1
timeout 1500 tools/profile_shapes.sh r04_a uhd fhd align > gpurun_out/r04c1/shapes.log 2>&1
tail -5 gpurun_out/r04c1/oct_uhd.txt
