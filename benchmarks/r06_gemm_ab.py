"""Round 6, experiment 2: A/B of the split-K projections of a Llama-3-8B decode layer at 64 rows.

  flags bit 0  K split taken from the XCD a workgroup runs on (1 / splits of the activation image per XCD's L2)
  flags bit 1  write-through fp32 partials
and a sweep of (waves per group, K splits) for qkv / o / down around the cost model's choice, with and without the XCD map.
Every launch streams a different layer's weights (16 layers = 7 GB: nothing waits in the 256 MiB infinity cache).
`chain` = the four projection calls of a layer back to back (attention left out), the way the step runs them.

    python benchmarks/r06_gemm_ab.py gpurun_out/r06_exp2_gemm_ab.json
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from sglang_amd import kernels as K, native  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
L, M, H, I, HQ, HKV, D = 16, 64, 4096, 14336, 32, 8, 128
NQ = (HQ + 2 * HKV) * D


def graph_time(fn, launches, reps=10, warm=6):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(warm):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * launches) * 1e3


def rnd(*shape):
    return (torch.randn(shape, device=dev) * 0.02).to(BF)


w_qkv = [rnd(NQ, H) for _ in range(L)]
w_o = [rnd(H, H) for _ in range(L)]
w_gu = [rnd(2 * I, H) for _ in range(L)]
w_dn = [rnd(H, I) for _ in range(L)]
norm_w = torch.ones(H, device=dev, dtype=BF)
xn = K.blocked_activation(M, H, dev)
xn.copy_(torch.randn(xn.shape, device=dev).to(BF))
a_in = torch.randn((M, H), device=dev).to(BF)
act_in = K.blocked_activation(M, I, dev)
act_in.copy_(torch.randn(act_in.shape, device=dev).to(BF))
res = torch.randn((M, H), device=dev).to(BF)
pos = torch.full((M,), 1024, dtype=torch.int64, device=dev)
loc = torch.arange(1, M + 1, dtype=torch.int64, device=dev)
cos_sin = torch.randn((2048, D), device=dev).to(BF)
kc = [torch.zeros((256, HKV, D), device=dev, dtype=BF) for _ in range(L)]
vc = [torch.zeros((256, HKV, D), device=dev, dtype=BF) for _ in range(L)]


def qkv(i, nw=None, s=None):
    return K.wstream_qkv_rope(xn, w_qkv[i], None, pos, cos_sin, HQ, HKV, D, kc[i], vc[i], loc, waves_per_group=nw, splits=s)


def oproj(i, nw=None, s=None):
    return K.wstream_gemm(a_in, w_o[i], epilogue="add_rmsnorm", residual=res, norm_weight=norm_w, eps=1e-5,
                          out_blocked=True, waves_per_group=nw, splits=s)


def gate_up(i):
    return K.wstream_gemm(xn, w_gu[i], epilogue="silu_and_mul", out_blocked=True)


def down(i, nw=None, s=None):
    return K.wstream_gemm(act_in, w_dn[i], epilogue="add_rmsnorm", residual=res, norm_weight=norm_w, eps=1e-5, out_blocked=True,
                          waves_per_group=nw, splits=s)


def set_flags(f):
    native.lib().sgl_amd_debug_wstream_flags(f)


out = {"what": "us per call (GEMM + combine pairs; gate_up alone), 64 rows, Llama-3-8B shapes, 16 layers' weights rotated", "flags": {}, "sweep": {}}
# bit-exactness of the switches
set_flags(0)
ref_q = qkv(0).clone(); ref_k = kc[0].clone()
r0 = res.clone()
ref_o = K.unblock(oproj(0)).clone(); res.copy_(r0)
ref_d = K.unblock(down(0)).clone(); res.copy_(r0)
for f in (1, 2, 3):
    set_flags(f)
    q = qkv(0)
    o = K.unblock(oproj(0)).clone(); res.copy_(r0)
    d = K.unblock(down(0)).clone(); res.copy_(r0)
    out.setdefault("bit_exact", {})[str(f)] = bool(torch.equal(q, ref_q) and torch.equal(kc[0], ref_k) and torch.equal(o, ref_o) and torch.equal(d, ref_d))
print("bit_exact", out["bit_exact"], flush=True)

for f in (0, 1, 2, 3):
    set_flags(f)
    rec = {}
    rec["qkv_rope"] = graph_time(lambda: [qkv(i) for i in range(L)], L)
    rec["o_norm"] = graph_time(lambda: [oproj(i) for i in range(L)], L)
    rec["gate_up"] = graph_time(lambda: [gate_up(i) for i in range(L)], L)
    rec["down_norm"] = graph_time(lambda: [down(i) for i in range(L)], L)
    rec["chain"] = graph_time(lambda: [(qkv(i), oproj(i), gate_up(i), down(i)) for i in range(L)], L)
    rec["sum"] = rec["qkv_rope"] + rec["o_norm"] + rec["gate_up"] + rec["down_norm"]
    out["flags"][str(f)] = rec
    print("flags", f, {k: round(v, 2) for k, v in rec.items()}, flush=True)

for f in (0, 1):
    set_flags(f)
    for name, fn, cfgs in (("qkv_rope", qkv, [(5, 3), (6, 4), (4, 4), (8, 4), (8, 2), (6, 2), (4, 8), (8, 8), (6, 8)]),
                           ("o_norm", oproj, [(4, 4), (8, 8), (8, 4), (4, 8), (8, 2), (4, 2), (8, 16)]),
                           ("down_norm", down, [(4, 4), (8, 8), (4, 8), (8, 4), (8, 2), (8, 16), (4, 16)])):
        for nw, s in cfgs:
            try:
                t = graph_time(lambda: [fn(i, nw, s) for i in range(L)], L)
            except Exception as e:      # noqa: BLE001
                t = None
                print(name, nw, s, "failed", str(e)[:100])
            out["sweep"][f"{name}_nw{nw}_s{s}_flags{f}"] = t
            print(f"{name} nw={nw} s={s} flags={f}: {t if t is None else round(t, 2)}", flush=True)
set_flags(0)
if len(sys.argv) > 1:
    Path(sys.argv[1]).parent.mkdir(exist_ok=True)
    Path(sys.argv[1]).write_text(json.dumps(out, indent=1))
