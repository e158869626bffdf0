"""Shared-prefix decode attention on the bench batch (4 groups x 16 requests, 896 shared + 192 private tokens):
bf16 rows vs OCP e4m3 rows (half the gathered bytes), per layer incl. the merge, graph-timed."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
Hq, Hkv, D, G, P, prefix, len_k, ctx = 32, 8, 128, 4, 16, 896, 1088, 1160
B = G * P
slots = B * 1200 + 4096


def graph_time(fn, reps=30):
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
    return e0.elapsed_time(e1) / reps * 1e3


k = (torch.randn((slots, Hkv, D), device=dev) * 0.5).to(BF)
v = (torch.randn((slots, Hkv, D), device=dev) * 0.5).to(BF)
r2t = torch.zeros((B + 1, ctx), dtype=torch.int32, device=dev)
perm = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
off = 0
for b in range(B):
    r2t[b + 1, :len_k] = perm[off: off + len_k]
    off += len_k
    r2t[b + 1, :prefix] = r2t[(b // P) * P + 1, :prefix]
pool = torch.arange(1, B + 1, device=dev)
seq = torch.full((B,), len_k, dtype=torch.int32, device=dev)
q = torch.randn((B, Hq, D), device=dev).to(BF)
ws = K.CascadeWorkspace(B, Hq, D, ctx, dev)
K.cascade_plan(ws, r2t, pool, seq, Hq, Hkv)
out = {}
res = {}
for name, fp8 in (("bf16", False), ("fp8_e4m3", True)):
    kc = torch.zeros((slots, Hkv, D), dtype=torch.uint8 if fp8 else BF, device=dev)
    vc = torch.zeros_like(kc)
    loc = torch.arange(slots, dtype=torch.int64, device=dev)
    K.store_kv_cache(k, v, kc, vc, loc, num_kv_heads=Hkv, head_dim=D, kv_fp8=fp8, k_scale=0.5, v_scale=0.5, page_size=1, hnd=False)
    o = torch.empty_like(q)
    kw = dict(kv_fp8=fp8, k_scale=0.5, v_scale=0.5) if fp8 else {}
    t = graph_time(lambda: K.cascade_decode_attention(ws, q, kc, vc, o, r2t, pool, seq, D ** -0.5, **kw))
    res[name] = o.float()
    out[name] = {"us_per_layer": t}
    print(name, round(t, 2), "us")
out["max_abs_diff_fp8_vs_bf16"] = float((res["bf16"] - res["fp8_e4m3"]).abs().max())
print(out)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp13_cascade_fp8.json").write_text(json.dumps(out, indent=1))
