"""Cascade vs plain paged decode attention on the bench's shared-prefix batch (hipGraph-timed)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

BF = torch.bfloat16
DEV = torch.device("cuda:0")


def timeit(fn, iters=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * reps) * 1e3


def slot_table(pattern, B, P, ctx, prefix, unique, width):
    """req_to_token of the bench batch under a slot pattern: `random` rows; `sequential`: every request's rows one after
    the other; `allocator`: what the token allocator hands out -- a group's prefix and every request's unique prompt rows
    contiguous (prefill), the decode rows of step s at base + s * B + b (one row per request and step)."""
    G = B // P
    slots = B * ctx + 64
    r2t = torch.zeros((B + 1, width), dtype=torch.int32, device=DEV)
    if pattern == "random":
        perm = (torch.randperm(slots - 1, device=DEV) + 1).to(torch.int32)
        for b in range(B):
            r2t[b + 1, :ctx] = perm[b * ctx:(b + 1) * ctx]
            r2t[b + 1, :prefix] = r2t[(b // P) * P + 1, :prefix]
    elif pattern == "sequential":
        for b in range(B):
            r2t[b + 1, :ctx] = torch.arange(1 + b * ctx, 1 + (b + 1) * ctx, device=DEV, dtype=torch.int32)
            r2t[b + 1, :prefix] = r2t[(b // P) * P + 1, :prefix]
    else:
        nxt = 1
        for g in range(G):                       # cold prefill: group leaders, prefix + unique each
            b = g * P
            r2t[b + 1, :prefix + unique] = torch.arange(nxt, nxt + prefix + unique, device=DEV, dtype=torch.int32)
            nxt += prefix + unique
        for b in range(B):                       # warm prefill: the others' unique rows
            if b % P:
                r2t[b + 1, :prefix] = r2t[(b // P) * P + 1, :prefix]
                r2t[b + 1, prefix:prefix + unique] = torch.arange(nxt, nxt + unique, device=DEV, dtype=torch.int32)
                nxt += unique
        steps = ctx - prefix - unique
        for b in range(B):
            r2t[b + 1, prefix + unique:ctx] = (nxt + torch.arange(steps, device=DEV) * B + b).to(torch.int32)
    return r2t, slots


def main():
    B, P, Hq, Hkv, D, prefix, unique = 64, 16, 32, 8, 128, 896, 128
    for pattern in ("random", "sequential", "allocator"):
        for ctx in (1025, 1088, 1151):
            r2t, slots = slot_table(pattern, B, P, ctx, prefix, unique, 1160)
            kc = torch.randn((slots, Hkv, D), device=DEV).to(BF)
            vc = torch.randn((slots, Hkv, D), device=DEV).to(BF)
            pool = torch.arange(1, B + 1, device=DEV)
            seq = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
            q = torch.randn((B, Hq, D), device=DEV).to(BF)
            out = torch.empty_like(q)
            out2 = torch.empty_like(q)
            ws = K.CascadeWorkspace(B, Hq, D, 1160, DEV)
            t_plan = timeit(lambda: K.cascade_plan(ws, r2t, pool, seq, Hq, Hkv))
            t_c = timeit(lambda: K.cascade_decode_attention(ws, q, kc, vc, out, r2t, pool, seq, D ** -0.5))
            t_p = timeit(lambda: K.decode_attention(q, kc, vc, out2, r2t, pool, seq, D ** -0.5))
            uniq = (4 * prefix + B * (ctx - prefix)) * 2 * Hkv * D * 2
            print(f"{pattern:10s} ctx {ctx}: plan {t_plan:.1f} us | cascade {t_c:.1f} us ({uniq / t_c / 1e3:.0f} GB/s unique) | plain {t_p:.1f} us | "
                  f"max diff {float((out.float() - out2.float()).abs().max()):.4f}")


if __name__ == "__main__":
    main()
