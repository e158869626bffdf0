#!/bin/bash
# round 6, experiment 1: LDS-DMA stream ceiling and the price of staging activations (benchmarks/r06_stream_ceiling.hip).
#   gpurun -- bash benchmarks/r06_exp1_stream_ceiling.sh
set -u
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 benchmarks/r06_stream_ceiling.hip -o /tmp/sc 2> gpurun_out/sc_build.log || { tail -5 gpurun_out/sc_build.log; exit 1; }
timeout 300 /tmp/sc gpurun_out/r06_exp1_stream_ceiling.json 2>&1 | tee gpurun_out/r06_exp1_stream_ceiling.txt
