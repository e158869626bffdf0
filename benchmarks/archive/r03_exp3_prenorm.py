"""(History: the kernel variant this script drives -- `wstream_gemm_prenorm` -- was measured and NOT kept; the script
documents how profiles/r03_exp3_prenorm.json was taken and does not run against the current library.)

One decoder layer's projections at the bench shapes (Llama-3-8B, M = 64), rotating weights, hipGraph-timed:
  chain A = today's launches: o (GEMM + combine_norm) -> gate_up+silu -> down (GEMM + combine_norm) -> qkv (GEMM + combine_rope)
  chain B = pre-norm launches: o (partials) -> [norm ⊕ gate_up+silu] -> down (partials) -> [norm ⊕ qkv] + combine_rope"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402
from sglang_amd.layers.rotary_embedding import get_rope  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
M, H, I, Hq, Hkv, D = 64, 4096, 14336, 32, 8, 128
L = 8
ws = [dict(o=torch.randn((H, H), device=dev).to(BF) * 0.02, gu=torch.randn((2 * I, H), device=dev).to(BF) * 0.02,
           down=torch.randn((H, I), device=dev).to(BF) * 0.02, qkv=torch.randn(((Hq + 2 * Hkv) * D, H), device=dev).to(BF) * 0.02)
      for _ in range(L)]
attn = torch.randn((M, H), device=dev).to(BF)
res = torch.randn((M, H), device=dev).to(BF)
nw_ = torch.ones(H, dtype=BF, device=dev)
rope = get_rope(D, D, 8192, 500000.0, True, None, BF, dev)
pos = torch.arange(M, device=dev) + 1000
loc = torch.arange(M, device=dev) + 1
kc = torch.zeros((256, Hkv, D), dtype=BF, device=dev)
vc = torch.zeros_like(kc)


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def chain_a():
    for w in ws:
        x = K.wstream_gemm(attn, w["o"], epilogue="add_rmsnorm", residual=res, norm_weight=nw_, eps=1e-5, out_blocked=True)
        a = K.wstream_gemm(x, w["gu"], epilogue="silu_and_mul", out_blocked=True)
        x = K.wstream_gemm(a, w["down"], epilogue="add_rmsnorm", residual=res, norm_weight=nw_, eps=1e-5, out_blocked=True)
        K.wstream_qkv_rope(x, w["qkv"], None, pos, rope.cos_sin_cache, Hq, Hkv, D, kc, vc, loc)


def chain_b():
    for w in ws:
        p = K.wstream_gemm_partials(attn, w["o"], buffer=0)
        a = K.wstream_gemm_prenorm(p, res, nw_, 1e-5, w["gu"], epilogue="silu_and_mul", out_blocked=True)
        p = K.wstream_gemm_partials(a, w["down"], buffer=1)
        K.wstream_qkv_rope(None, w["qkv"], None, pos, rope.cos_sin_cache, Hq, Hkv, D, kc, vc, loc, prenorm=(p, res, nw_, 1e-5))


def parts():
    out = {}
    out["o+norm"] = graph_time(lambda: [K.wstream_gemm(attn, w["o"], epilogue="add_rmsnorm", residual=res, norm_weight=nw_, eps=1e-5, out_blocked=True) for w in ws]) / L
    out["o_partials"] = graph_time(lambda: [K.wstream_gemm_partials(attn, w["o"]) for w in ws]) / L
    x = K.blocked_activation(M, H, dev)
    out["gate_up"] = graph_time(lambda: [K.wstream_gemm(x, w["gu"], epilogue="silu_and_mul", out_blocked=True) for w in ws]) / L
    p = K.wstream_gemm_partials(attn, ws[0]["o"], buffer=0)
    out["norm+gate_up"] = graph_time(lambda: [K.wstream_gemm_prenorm(p, res, nw_, 1e-5, w["gu"], epilogue="silu_and_mul", out_blocked=True) for w in ws]) / L
    out["qkv+rope"] = graph_time(lambda: [K.wstream_qkv_rope(x, w["qkv"], None, pos, rope.cos_sin_cache, Hq, Hkv, D, kc, vc, loc) for w in ws]) / L
    p1 = K.wstream_gemm_partials(K.blocked_activation(M, I, dev), ws[0]["down"], buffer=1)
    out["norm+qkv+rope"] = graph_time(lambda: [K.wstream_qkv_rope(None, w["qkv"], None, pos, rope.cos_sin_cache, Hq, Hkv, D, kc, vc, loc, prenorm=(p1, res, nw_, 1e-5)) for w in ws]) / L
    return out


r = {"chain_a_us_per_layer": graph_time(chain_a) / L, "chain_b_us_per_layer": graph_time(chain_b) / L, "parts_us": parts(),
     "timed_out": K.prenorm_timed_out(dev)}
print(json.dumps(r, indent=1))
(Path(__file__).resolve().parent.parent / "gpurun_out" / "r03_exp3_prenorm.json").write_text(json.dumps(r, indent=1))
