#!/bin/bash
# prototype of the persistent decode layer's weight stream (benchmarks/persistent_stream_proto.hip): launch boundary against an
# in-launch grid barrier, with and without the next matrix prefetched across it.   gpurun -- bash benchmarks/r05_exp3_persistent_stream.sh
set -u
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 benchmarks/persistent_stream_proto.hip -o /tmp/proto 2> gpurun_out/proto_build.log || { tail -5 gpurun_out/proto_build.log; exit 1; }
timeout 120 /tmp/proto gpurun_out/r05_exp3_persistent_stream.json 2>&1 | tail -14
