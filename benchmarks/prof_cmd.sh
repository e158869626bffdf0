#!/bin/bash
# rocprofv3 kernel stats of an arbitrary command: benchmarks/prof_cmd.sh <out-name> <cmd...>
set -u
NAME=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
rm -rf /tmp/prof_$NAME
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o run -- "$@" > $REPO/gpurun_out/prof_$NAME.log 2>&1
cd $REPO && python benchmarks/summarize_rocprof.py /tmp/prof_$NAME gpurun_out/prof_$NAME.txt 40 | cut -c1-180 | head -${TOPN:-14}
