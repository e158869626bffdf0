"""The bench job laid out for rocprofv3 --pmc passes: every kernel is dispatched eagerly (no hipGraph) and the
phases are separated by a sentinel launch (`mfma_probe_kernel`, used by nothing else), so that
benchmarks/summarize_pmc_phases.py can attribute counters per phase and per kernel:

    warmup | S | prefill_cold | S | prefill_warm | S | decode x N | S

    rocprofv3 --pmc FETCH_SIZE -f csv -d /tmp/pmc_FETCH -o run -- python benchmarks/pmc_workload.py
"""
import argparse
import dataclasses
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from bench import build_prompts  # noqa: E402
from sglang_amd import kernels  # noqa: E402
from sglang_amd.harness.engine import Engine, ModelRunner, Req  # noqa: E402
from sglang_amd.harness.models import CONFIGS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama-3-8b")
ap.add_argument("--layers", type=int, default=0)
ap.add_argument("--decode-steps", type=int, default=2)
ap.add_argument("--groups", type=int, default=4)
ap.add_argument("--per-group", type=int, default=16)
ap.add_argument("--prefix", type=int, default=896)
ap.add_argument("--unique", type=int, default=128)
args = ap.parse_args()

dev = torch.device("cuda:0")
cfg = CONFIGS[args.model]
if args.layers:
    cfg = dataclasses.replace(cfg, num_hidden_layers=args.layers)
G, P = args.groups, args.per_group
B = G * P
in_len = args.prefix + args.unique
runner = ModelRunner(cfg, max_total_tokens=B * (in_len + 64) + 4096, max_running_requests=B, max_context_len=in_len + 72,
                     device=dev, use_graph=False)
eng = Engine(runner)
prompts = build_prompts(cfg, G, P, args.prefix, args.unique)
pa = torch.zeros((16, 32), dtype=torch.bfloat16, device=dev)
pb = torch.zeros((32, 16), dtype=torch.bfloat16, device=dev)


def sentinel():
    torch.cuda.synchronize()
    kernels.probe_mfma_16x16x32(pa, pb)
    torch.cuda.synchronize()


def job(mark: bool):
    runner.tree_cache.reset()
    runner.token_to_kv_pool_allocator.clear()
    runner.req_to_token_pool.clear()
    reqs = [Req(g * P + p, prompts[g][p], 64) for g in range(G) for p in range(P)]
    leaders = [q for q in reqs if q.rid % P == 0]
    rest = [q for q in reqs if q.rid % P]
    if mark:
        sentinel()
    eng.prefill(leaders)
    if mark:
        sentinel()
    eng.prefill(rest)
    if mark:
        sentinel()
    for _ in range(args.decode_steps):
        eng.decode_step()
    if mark:
        sentinel()
    eng.finish(list(eng.running))


job(False)      # lazy library initialisation, allocator warm-up
job(True)
torch.cuda.synchronize()
print(f"pmc_workload done: {cfg.name} L={cfg.num_hidden_layers} B={B} decode_steps={args.decode_steps}")
