#!/usr/bin/env python
"""bench.py -- the RadixAttention serving hot path on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete pass of the hot path over the synthetic shared-prefix
batch of BASELINE.json `configs[1]` (Llama-3-8B, bf16, 1024-in / 128-out):
  P-cold  prefill of the G group leaders (1024 new tokens each, empty radix tree)
  P-warm  prefill of the other B-G requests (896 tokens hit the radix cache, 128 new)
  D       127 decode steps at batch B (hipGraph replay), greedy, ignore_eos
followed by cache_finished_req for every request.  The cache is reset between
steps so every step does identical work.  value = output tokens / s over the K
timed steps (max over ranks); at N > 1 the model runs TP=N over RCCL with the
batch scaled to 64*N requests (weak scaling).

The JSON line also carries `roofline` (the dominant hand-written kernel: the
weight-streaming gate_up GEMM, HBM bound, measured live with HIP events over the
model's own 32 layers), `attention_roofline` (cascade / plain decode attention
on the workload's slot pattern), `step_roofline`
(whole decode step vs 8 TB/s), `prefill_mfma_frac`, p50 TTFT,
and `cpu_baseline` (the CPU oracle = reference torch-native path, on a bounded
sample of the same workload, rank 0 / N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import statistics
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0     # bf16 dense


def build_prompts(cfg, groups, per_group, prefix, unique, seed=1):
    """gen_prompt-style synthetic ids (benchmark/datasets/common.py:79-83): uniform in [0, vocab)."""
    rnd = random.Random(seed)
    prompts = []
    for _ in range(groups):
        sys_p = [rnd.randrange(cfg.vocab_size) for _ in range(prefix)]
        prompts.append([sys_p + [rnd.randrange(cfg.vocab_size) for _ in range(unique)] for _ in range(per_group)])
    return prompts


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc summaries (separate passes for
    FETCH_SIZE and WRITE_SIZE, KiB per dispatch; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of a
    wide coalesced streaming read, so it is doubled).  None when the summaries are not there."""
    vals = {}
    for ctr, fn in (("FETCH_SIZE", "r01_gemm_pmc_fetch.txt"), ("WRITE_SIZE", "r01_gemm_pmc_write.txt")):
        f = ROOT / "profiles" / fn
        if not f.exists():
            return None
        for line in f.read_text().splitlines():
            if kernel_substr in line and ctr in line:
                vals[ctr] = float(line.split("avg")[1].split()[0])
    if len(vals) != 2:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def p_lin(cfg):
    """Matmul weights touched per token, excluding embedding / lm_head."""
    H, D = cfg.hidden_size, cfg.head_dim
    qkv = H * (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * D
    o = cfg.num_attention_heads * D * H
    mlp = 3 * H * cfg.intermediate_size
    return cfg.num_hidden_layers * (qkv + o + mlp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--groups", type=int, default=4, help="prompt groups per GPU")
    ap.add_argument("--per-group", type=int, default=16)
    ap.add_argument("--prefix", type=int, default=896)
    ap.add_argument("--unique", type=int, default=128)
    ap.add_argument("--out", type=int, default=128)
    ap.add_argument("--page-size", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback path")

    from sglang_amd.distributed import parallel_state as ps
    from sglang_amd.harness.engine import Engine, ModelRunner, Req
    from sglang_amd.harness.models import CONFIGS
    import torch.distributed as dist

    ps.init_distributed_environment()
    world = ps.get_tensor_model_parallel_world_size()
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = CONFIGS[args.model]

    G = args.groups * world          # weak scaling: the batch grows with the TP degree
    P = args.per_group
    B = G * P
    in_len = args.prefix + args.unique
    ctx = in_len + args.out + 8
    prompts = build_prompts(cfg, G, P, args.prefix, args.unique)

    runner = ModelRunner(cfg, max_total_tokens=B * (in_len + args.out) + 4096, max_running_requests=B,
                         max_context_len=ctx, page_size=args.page_size, device=dev, use_graph=not args.no_graph,
                         graph_max_bs=B)
    eng = Engine(runner)

    def sync():
        torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    phase_times = []
    ttfts = []

    def job(record: bool):
        # identical work every step: drop the previous step's tree / slots
        runner.tree_cache.reset()
        runner.token_to_kv_pool_allocator.clear()
        runner.req_to_token_pool.clear()
        t0 = time.perf_counter()
        rid = 0
        leaders, rest = [], []
        for g in range(G):
            for p in range(P):
                q = Req(rid, prompts[g][p], args.out)
                q.t_arrive = t0
                (leaders if p == 0 else rest).append(q)
                rid += 1
        eng.prefill(leaders)
        sync(); t1 = time.perf_counter()
        if rest:
            eng.prefill(rest)
        sync(); t2 = time.perf_counter()
        for _ in range(args.out - 1):
            eng.decode_step()
            eng.flush_decode_outputs(lag=1)   # per-step token hand-off, one step behind the launch (overlap scheduling)
        sync(); t3 = time.perf_counter()
        reqs = list(eng.running)
        hit = sum(q.cached_tokens for q in reqs)
        eng.finish(reqs)
        t4 = time.perf_counter()
        if record:
            phase_times.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
            ttfts.extend(q.t_first_token - q.t_arrive for q in reqs)
        assert all(len(q.output_ids) == args.out for q in reqs)
        return hit

    for _ in range(args.warmup):
        job(False)
    sync(); barrier()
    t_start = time.perf_counter()
    hit_tokens = 0
    for _ in range(args.steps):
        hit_tokens = job(True)
    sync(); barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out_tokens = B * args.out * args.steps
    value = out_tokens / elapsed
    cold = statistics.mean(p[0] for p in phase_times)
    warm = statistics.mean(p[1] for p in phase_times)
    dec = statistics.mean(p[2] for p in phase_times)
    t_decode_step = dec / max(1, args.out - 1)

    # ---- rooflines ------------------------------------------------
    L, Hq, Hkv, D = cfg.num_hidden_layers, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    kv_row = 2 * L * Hkv * D * 2                      # bytes per cached token, all layers (131072 for 8B)
    plin = p_lin(cfg)
    w_act = (plin + cfg.hidden_size * cfg.vocab_size) * 2 / world
    mean_len = in_len + args.out / 2
    kv_unique = (G * args.prefix + B * (mean_len - args.prefix)) * kv_row / world
    kv_nodedup = B * mean_len * kv_row / world
    kv_write = B * kv_row / world
    step_bytes = w_act + kv_unique + kv_write
    step_roofline = dict(bound="hbm", achieved=step_bytes / t_decode_step / 1e9, peak=HBM_PEAK_GBPS, unit="GB/s",
                         frac=step_bytes / t_decode_step / 1e9 / HBM_PEAK_GBPS, traffic=None,
                         bytes_per_step=step_bytes, kv_bytes_no_dedup=kv_nodedup, ms_per_decode_step=t_decode_step * 1e3)
    pair = 4 * L * Hq * D
    flops_cold = G * (2 * in_len * plin + pair * (in_len * (in_len + 1) / 2) + 2 * cfg.hidden_size * cfg.vocab_size)
    flops_warm = (B - G) * (2 * args.unique * plin + pair * (args.unique * args.prefix + args.unique * (args.unique + 1) / 2)
                            + 2 * cfg.hidden_size * cfg.vocab_size)
    prefill_tflops = (flops_cold + flops_warm) / world / (cold + warm) / 1e12

    result = {
        "metric": "output tokens/s + p50 TTFT, Llama-3-8B TP=1 shared-prefix batch; 70B TP=8",
        "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{cfg.name} shared-prefix batch: {G} groups x {P} prompts, {args.prefix} shared + "
                               f"{args.unique} unique in, {args.out} out, greedy, page_size {args.page_size}",
                   "model": cfg.name, "global_batch": B, "seq_len": in_len, "parallelism": f"tp{world}",
                   "decode": "hipGraph" if runner.graph_runner is not None else "eager"},
        "ttft_p50_ms": statistics.median(ttfts) * 1e3,
        "phase_ms": {"prefill_cold": cold * 1e3, "prefill_warm": warm * 1e3, "decode": dec * 1e3},
        "decode_tokens_per_s": B / t_decode_step,
        "radix_hit_tokens": hit_tokens,
        "step_roofline": step_roofline,
        "prefill_mfma": {"achieved": prefill_tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": prefill_tflops / MFMA_PEAK_TFLOPS},
    }

    # ---- dominant hand-written kernels, measured live with HIP events on torch's stream ----
    if rank == 0 and not args.no_kernel_roofline:
        try:
            from sglang_amd import kernels as K

            def graph_time(fn, launches, reps=5):
                """Average duration of one launch: `fn` (which enqueues `launches` kernels) captured into a
                hipGraph, replayed `reps` times between two HIP events on the current stream."""
                fn()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    fn()
                g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / (reps * launches) * 1e-3

            # (1) the kernel with the largest share of the step: the weight-streaming GEMM of gate_up_proj
            # (fused silu_and_mul epilogue).  One launch per layer over the model's OWN weights, so every
            # launch streams a different 235 MB from HBM (nothing is left in the 256 MiB infinity cache).
            mlps = [layer.mlp for layer in runner.model.layers if hasattr(layer.mlp, "gate_up_proj")]
            Mg = min(B, 64)               # the weight-streaming design point (larger per-rank batches: DESIGN.md)
            if mlps and K.wstream_preferred(Mg, *mlps[0].gate_up_proj.weight.shape):
                xg = torch.randn((Mg, cfg.hidden_size), device=dev).to(torch.bfloat16)
                wN, wK = mlps[0].gate_up_proj.weight.shape
                t_g = graph_time(lambda: [m.gate_up_act(xg) for m in mlps], len(mlps))
                alg = wN * wK * 2 + Mg * wK * 2 + Mg * (wN // 2) * 2      # weights once + activations in + out
                nw_s = K.choose_wstream_config(Mg, wN, wK, True, True)
                traffic = pmc_traffic_bytes("wstream_gemm_kernel") if (Mg, wN, wK) == (64, 28672, 4096) else None
                result["roofline"] = {"bound": "hbm", "achieved": alg / t_g / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                      "frac": alg / t_g / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                                      "traffic_source": "profiles/r01_gemm_pmc_{fetch,write}.txt: 2 x FETCH_SIZE (gfx950 "
                                                        "counts 64 B per 128 B request) + WRITE_SIZE, KiB per dispatch",
                                      "kernel": "wstream_gemm_kernel<4,4,2> (gate_up_proj + silu_and_mul)",
                                      "us_per_launch": t_g * 1e6, "bytes_per_launch": alg,
                                      "shape": {"M": Mg, "N": wN, "K": wK, "waves_per_group": nw_s[0], "k_splits": nw_s[1]},
                                      "share_of_decode_step": len(mlps) * t_g / t_decode_step}

            # (2) decode attention over the workload's own slot pattern (shared prefix rows + private rows)
            from sglang_amd.layers.attention.hip_backend import choose_num_splits

            Hq_r, Hkv_r = runner.num_attention_heads_per_rank, runner.num_kv_heads_per_rank
            len_k = in_len + args.out // 2
            r2t = runner.req_to_token_pool.req_to_token
            perm = (torch.randperm(runner.token_to_kv_pool.size - 1, device=dev) + 1).to(torch.int32)
            off = 0
            for b in range(B):
                r2t[b + 1, :len_k] = perm[off: off + len_k]
                off += len_k
                leader = (b // P) * P
                r2t[b + 1, :args.prefix] = r2t[leader + 1, :args.prefix]
            pool_idx = torch.arange(1, B + 1, device=dev)
            seq = torch.full((B,), len_k, dtype=torch.int32, device=dev)
            q = torch.randn((B, Hq_r, D), device=dev).to(torch.bfloat16)
            o = torch.empty_like(q)
            kc, vc = runner.token_to_kv_pool.get_key_buffer(0), runner.token_to_kv_pool.get_value_buffer(0)
            kc.normal_(); vc.normal_()
            row_bytes = 2 * Hkv_r * D * 2
            att = {}
            if D in (64, 128) and B >= 2:
                cws = K.CascadeWorkspace(B, Hq_r, D, ctx, dev)
                K.cascade_plan(cws, r2t, pool_idx, seq, Hq_r, Hkv_r)
                t_c = graph_time(lambda: K.cascade_decode_attention(cws, q, kc, vc, o, r2t, pool_idx, seq, D ** -0.5), 1, reps=20)
                uniq = (G * args.prefix + B * (len_k - args.prefix)) * row_bytes
                att["cascade"] = {"kernel": "cascade_chunk_kernel + cascade_merge2_kernel", "us_per_layer": t_c * 1e6,
                                  "bytes_unique": uniq, "achieved": uniq / t_c / 1e9, "frac": uniq / t_c / 1e9 / HBM_PEAK_GBPS}
            splits = choose_num_splits(B, Hkv_r, Hq_r // Hkv_r, len_k)
            ws = K.decode_workspace(B, Hq_r, D, splits, dev) if splits > 1 else (None, None)
            t_k = graph_time(lambda: K.decode_attention(q, kc, vc, o, r2t, pool_idx, seq, D ** -0.5, splits, ws[0], ws[1]), 1, reps=20)
            alg = B * len_k * row_bytes     # len * (2*H_kv*D*2 B) per request and layer, no dedup
            att["plain"] = {"kernel": "decode_stage1_kernel", "us_per_layer": t_k * 1e6, "bytes_no_dedup": alg,
                            "achieved": alg / t_k / 1e9, "frac": alg / t_k / 1e9 / HBM_PEAK_GBPS}
            result["attention_roofline"] = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                            "shape": {"B": B, "Hq": Hq_r, "Hkv": Hkv_r, "D": D, "kv_len": len_k,
                                                      "shared_prefix": args.prefix, "groups": G}, **att}
        except Exception as e:      # the measured line must survive a failure of these side measurements
            result.setdefault("roofline", {"error": f"{type(e).__name__}: {e}"})

    # ---- CPU baseline: the oracle (reference torch-native path) on host cores --------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.model import OracleLM, weights_from_product_model

        # CPU bf16 GEMMs of an 8B model run at a few GFLOP/s, so the sample has to be tiny to stay
        # inside the time budget: ONE request, 3 shared + 1 unique prompt tokens, then as many
        # greedy decode steps as the budget allows (estimated from the first forward).
        w = weights_from_product_model(runner.model)
        nb, npre, nuni = 1, 3, 1
        sample = [prompts[0][i][:npre] + prompts[0][i][args.prefix:args.prefix + nuni] for i in range(nb)]
        t0 = time.perf_counter()
        OracleLM(cfg, w, num_slots=64, max_ctx=64).generate(sample, 1)
        t_first = time.perf_counter() - t0
        n_tok, t_tot, n_out = nb, t_first, 1
        remaining = args.cpu_budget_s - t_first
        per_step = t_first / (npre + nuni)            # a decode step costs about one prompt token
        steps = int(min(8, remaining / max(per_step, 1e-3) - (npre + nuni)))
        if steps >= 1:
            t0 = time.perf_counter()
            OracleLM(cfg, w, num_slots=64, max_ctx=64).generate(sample, 1 + steps)
            t_tot = time.perf_counter() - t0
            n_tok, n_out = nb * (1 + steps), 1 + steps
        result["cpu_baseline"] = {"value": n_tok / t_tot, "unit": "tokens/s", "cores": torch.get_num_threads(),
                                  "kind": "port",
                                  "sample": f"{cfg.name} oracle (CPU torch-native restatement of the reference path), "
                                            f"{nb} request x ({npre} shared + {nuni} unique) tokens in, {n_out} out, "
                                            f"greedy, bf16, the same synthetic weights copied from the GPU; "
                                            f"{t_tot:.1f} s wall",
                                  "host_cpu_count": os.cpu_count()}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        ps.destroy()


if __name__ == "__main__":
    main()
