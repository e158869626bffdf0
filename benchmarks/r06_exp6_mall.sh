#!/bin/bash
# round 6, experiment 6: weight streams after a prefetch through the infinity cache (benchmarks/r06_stream_ceiling.hip, "mall" mode)
set -e
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 benchmarks/r06_stream_ceiling.hip -o /tmp/r06_sc
timeout 300 /tmp/r06_sc gpurun_out/r06_exp6_mall.json mall | tee gpurun_out/r06_exp6_mall.txt
