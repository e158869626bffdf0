#!/bin/bash
# prototype: KV-row gather rate against the contiguous piece read per token (benchmarks/gather_proto.hip).   gpurun -- bash benchmarks/r05_exp4_gather.sh
set -u
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 benchmarks/gather_proto.hip -o /tmp/gather 2> gpurun_out/gather_build.log || { tail -5 gpurun_out/gather_build.log; exit 1; }
timeout 120 /tmp/gather gpurun_out/r05_exp4_gather.json 2>&1 | grep -v "^{"
