# the driver's exact multi-GPU command at the real job's size (default flags), two ranks time-slicing GPU 0
export SGLANG_AMD_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 SGLANG_USE_AITER=0
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 > gpurun_out/r06_dryrun_tp2_one_gpu.json 2> gpurun_out/r06_dryrun_tp2.err
echo rc=$?
python - <<'P'
import json
ls=[l for l in open('gpurun_out/r06_dryrun_tp2_one_gpu.json').read().splitlines() if l.startswith('{')]
d=json.loads(ls[-1]); print(len(ls), d['n_gpus'], round(d['value'],1), d['config']['parallelism'][:90], d['config']['global_batch'])
P
tail -3 gpurun_out/r06_dryrun_tp2.err | cut -c1-300
