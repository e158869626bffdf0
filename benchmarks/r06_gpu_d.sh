# rank shapes of TP 8 on one GPU (Llama-3-8B): where the per-rank decode step goes at 512 rows
F="--no-cpu-baseline --no-kernel-roofline --no-parity --no-reference-scheduler"
TOPN=45 bash benchmarks/prof_cmd.sh r06_rankof8 python $(pwd)/bench.py --rank-of 8 --steps 1 --warmup 1 $F
cp gpurun_out/prof_r06_rankof8.txt gpurun_out/r06_rankof8_kernel_stats.txt
tail -2 gpurun_out/prof_r06_rankof8.log | cut -c1-1500
