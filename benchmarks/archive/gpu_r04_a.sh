#!/bin/bash
# round 4, call A: Llama-3-70B WHOLE at TP=1 on one MI355X (141 GB of bf16 weights in 288 GB): bench line with the
# teacher-forced 80-layer parity leg + per-layer parity, then the kernel stats of one step
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1100 python bench.py --model llama-3-70b --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_line_70b_tp1.json 2> gpurun_out/bench_70b.log
echo "rc=$?"; head -c 900 gpurun_out/r04_bench_line_70b_tp1.json; echo; tail -3 gpurun_out/bench_70b.log | cut -c1-300
bash benchmarks/gpu_prof.sh --model llama-3-70b --no-parity --no-kernel-roofline > gpurun_out/prof_70b.log 2>&1
mv gpurun_out/kernel_stats.txt gpurun_out/r04_70b_tp1_kernel_stats.txt
head -25 gpurun_out/r04_70b_tp1_kernel_stats.txt | cut -c1-180
free -g | head -2
