"""Per-kernel microbenchmarks on one MI355X (hipEvent timing on torch's stream).

    python benchmarks/micro.py [--json gpurun_out/micro.json]

Reports achieved algorithmic GB/s (or TFLOP/s) per kernel at Llama-3-8B shapes.
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

BF = torch.bfloat16
DEV = torch.device("cuda:0")


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def bench_decode(B, Hq, Hkv, D, ctx, splits, shared_prefix=0, groups=1, flags=0):
    slots = B * ctx + 64
    kc = torch.randn((slots, Hkv, D), device=DEV).to(BF)
    vc = torch.randn((slots, Hkv, D), device=DEV).to(BF)
    r2t = torch.zeros((B + 1, ctx + 8), dtype=torch.int32, device=DEV)
    perm = (torch.randperm(slots - 1, device=DEV) + 1).to(torch.int32)
    for b in range(B):
        r2t[b + 1, :ctx] = perm[b * ctx:(b + 1) * ctx]
        if shared_prefix:
            leader = (b // (B // groups)) * (B // groups)
            r2t[b + 1, :shared_prefix] = r2t[leader + 1, :shared_prefix]
    pool = torch.arange(1, B + 1, device=DEV)
    seq = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
    q = torch.randn((B, Hq, D), device=DEV).to(BF)
    out = torch.empty_like(q)
    ws = K.decode_workspace(B, Hq, D, splits, DEV) if splits > 1 else (None, None)
    t = timeit(lambda: K.decode_attention(q, kc, vc, out, r2t, pool, seq, D ** -0.5, splits, ws[0], ws[1], flags=flags))
    kv_bytes = B * ctx * 2 * Hkv * D * 2
    return dict(kernel="decode_attention", B=B, Hq=Hq, Hkv=Hkv, D=D, ctx=ctx, splits=splits,
                shared_prefix=shared_prefix, flags=flags, us=t * 1e6, GBps=kv_bytes / t / 1e9)


def bench_extend(B, Hq, Hkv, D, prefix, ext):
    seq_len = prefix + ext
    slots = B * seq_len + 64
    kc = torch.randn((slots, Hkv, D), device=DEV).to(BF)
    vc = torch.randn((slots, Hkv, D), device=DEV).to(BF)
    r2t = torch.zeros((B + 1, seq_len + 8), dtype=torch.int32, device=DEV)
    perm = (torch.randperm(slots - 1, device=DEV) + 1).to(torch.int32)
    for b in range(B):
        r2t[b + 1, :seq_len] = perm[b * seq_len:(b + 1) * seq_len]
    pool = torch.arange(1, B + 1, device=DEV)
    seq = torch.full((B,), seq_len, dtype=torch.int32, device=DEV)
    pre = torch.full((B,), prefix, dtype=torch.int32, device=DEV)
    qo = torch.arange(0, (B + 1) * ext, ext, dtype=torch.int32, device=DEV)
    q = torch.randn((B * ext, Hq, D), device=DEV).to(BF)
    out = torch.empty_like(q)
    t = timeit(lambda: K.extend_attention(q, out, kc, vc, r2t, pool, seq, pre, qo, ext, D ** -0.5, True))
    flops = B * 4 * Hq * D * (ext * prefix + ext * (ext + 1) / 2)
    return dict(kernel="extend_attention", B=B, Hq=Hq, Hkv=Hkv, D=D, prefix=prefix, ext=ext, us=t * 1e6,
                TFLOPs=flops / t / 1e12)


def bench_elementwise(T, hidden=4096, inter=14336):
    res = []
    x = torch.randn((T, hidden), device=DEV).to(BF)
    r = torch.randn((T, hidden), device=DEV).to(BF)
    w = torch.ones(hidden, device=DEV, dtype=BF)
    t = timeit(lambda: K.fused_add_rmsnorm(x, r, w, 1e-5))
    res.append(dict(kernel="fused_add_rmsnorm", T=T, us=t * 1e6, GBps=T * hidden * 2 * 4 / t / 1e9))
    out = torch.empty_like(x)
    t = timeit(lambda: K.rmsnorm(x, w, 1e-5, out))
    res.append(dict(kernel="rmsnorm", T=T, us=t * 1e6, GBps=T * hidden * 2 * 2 / t / 1e9))
    gu = torch.randn((T, 2 * inter), device=DEV).to(BF)
    o = torch.empty((T, inter), device=DEV, dtype=BF)
    t = timeit(lambda: K.silu_and_mul(gu, o))
    res.append(dict(kernel="silu_and_mul", T=T, us=t * 1e6, GBps=T * inter * 2 * 3 / t / 1e9))
    Hq, Hk, D = 32, 8, 128
    q = torch.randn((T, Hq * D), device=DEV).to(BF)
    k = torch.randn((T, Hk * D), device=DEV).to(BF)
    v = torch.randn((T, Hk * D), device=DEV).to(BF)
    cache = torch.randn((8192, D), device=DEV).to(BF)
    pos = torch.randint(0, 8192, (T,), device=DEV)
    kc = torch.zeros((T + 8, Hk, D), device=DEV, dtype=BF)
    vc = torch.zeros((T + 8, Hk, D), device=DEV, dtype=BF)
    loc = torch.randperm(T, device=DEV) + 1
    t = timeit(lambda: K.rotary_embedding(pos, q, k, D, cache, True, value=v, k_cache=kc, v_cache=vc, cache_loc=loc))
    res.append(dict(kernel="rope+kv_store", T=T, us=t * 1e6,
                    GBps=T * ((Hq + Hk) * D * 4 + Hk * D * 2 * 3) / t / 1e9))
    logits = torch.randn((min(T, 256), 128256), device=DEV)
    t = timeit(lambda: K.argmax(logits))
    res.append(dict(kernel="argmax", B=logits.shape[0], us=t * 1e6, GBps=logits.numel() * 4 / t / 1e9))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    res = []
    for splits in (1, 2, 4):
        res.append(bench_decode(64, 32, 8, 128, 1088, splits))
    res.append(bench_decode(64, 32, 8, 128, 1088, 2, flags=1))
    res.append(bench_decode(64, 32, 8, 128, 1088, 2, shared_prefix=896, groups=4))
    res.append(bench_decode(256, 32, 8, 128, 1088, 1))
    res.append(bench_decode(1, 32, 8, 128, 8192, 32))
    res.append(bench_decode(64, 8, 1, 128, 1088, 8))
    res.append(bench_extend(4, 32, 8, 128, 0, 1024))
    res.append(bench_extend(60, 32, 8, 128, 896, 128))
    res.append(bench_extend(1, 32, 8, 128, 0, 8192))
    res += bench_elementwise(64)
    res += bench_elementwise(4096)
    for r in res:
        print(json.dumps(r))
    if args.json:
        Path(args.json).parent.mkdir(parents=True, exist_ok=True)
        Path(args.json).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
