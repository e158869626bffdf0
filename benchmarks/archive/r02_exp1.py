"""Round-2 kernel A/B on one MI355X: cascade decode attention (two-image items) and extend attention with the
alternating stager halves (SGL_AMD_EXTEND_DEEP=0/1), on the bench workload's shapes.  hipGraph replays between events."""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out = {}
Hq, Hkv, D = 32, 8, 128
G, P, prefix = 4, 16, 896
B = G * P
slots = B * 1200 + 4096
kc = torch.randn((slots, Hkv, D), device=dev).to(BF)
vc = torch.randn((slots, Hkv, D), device=dev).to(BF)
for len_k in (1030, 1088, 1150):
    ctx = 1160
    r2t = torch.zeros((B + 1, ctx), dtype=torch.int32, device=dev)
    perm = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
    off = 0
    for b in range(B):
        r2t[b + 1, :len_k] = perm[off: off + len_k]; off += len_k
        r2t[b + 1, :prefix] = r2t[(b // P) * P + 1, :prefix]
    pool = torch.arange(1, B + 1, device=dev)
    seq = torch.full((B,), len_k, dtype=torch.int32, device=dev)
    q = torch.randn((B, Hq, D), device=dev).to(BF)
    o = torch.empty_like(q); o2 = torch.empty_like(q)
    ws = K.CascadeWorkspace(B, Hq, D, ctx, dev)
    K.cascade_plan(ws, r2t, pool, seq, Hq, Hkv)
    t = graph_time(lambda: K.cascade_decode_attention(ws, q, kc, vc, o, r2t, pool, seq, D ** -0.5))
    K.decode_attention(q, kc, vc, o2, r2t, pool, seq, D ** -0.5)
    err = float((o.float() - o2.float()).abs().max())
    uniq = (G * prefix + B * (len_k - prefix)) * 2 * Hkv * D * 2
    out[f"cascade_len{len_k}"] = {"us": t, "GBps_unique": uniq / t / 1e3, "frac_hbm": uniq / t / 1e3 / 8000, "max_err_vs_plain": err,
                                  "items": int(ws.plan[0])}
    print(f"cascade len {len_k}: {t:.1f} us, {uniq / t / 1e3:.0f} GB/s unique, err vs plain {err:.4f}, items {int(ws.plan[0])}")

# extend attention, cold (4 x 1024 causal) and warm (60 x 128 over 896)
ctx = 1160
r2t = torch.zeros((B + 1, ctx), dtype=torch.int32, device=dev)
perm = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
off = 0
for b in range(B):
    r2t[b + 1, :1024] = perm[off: off + 1024]; off += 1024
    r2t[b + 1, :prefix] = r2t[(b // P) * P + 1, :prefix]
for name, nreq, pre, e in (("cold", G, 0, 1024), ("warm", B - G, prefix, 128), ("long", 2, 0, 8192)):
    if name == "long":
        r2t = torch.zeros((3, 8200), dtype=torch.int32, device=dev)
        r2t[1, :8192] = perm[:8192]; r2t[2, :8192] = perm[8192:16384]
    T = nreq * e
    qx = torch.randn((T, Hq, D), device=dev).to(BF)
    seq_x = torch.full((nreq,), pre + e, dtype=torch.int32, device=dev)
    pre_x = torch.full((nreq,), pre, dtype=torch.int32, device=dev)
    qo = (torch.arange(nreq + 1, device=dev) * e).to(torch.int32)
    pool_x = torch.arange(1, nreq + 1, device=dev)
    res = {}
    outs = {}
    for deep in ("0", "1"):
        os.environ["SGL_AMD_EXTEND_DEEP"] = deep
        ox = torch.empty_like(qx)
        t = graph_time(lambda: K.extend_attention(qx, ox, kc, vc, r2t, pool_x, seq_x, pre_x, qo, e, D ** -0.5, True), reps=10)
        fl = nreq * 4 * Hq * D * (e * pre + e * (e + 1) / 2)
        res[f"deep{deep}"] = {"us": t, "tflops": fl / t / 1e6}
        outs[deep] = ox.clone()
    res["max_diff_deep_vs_flat"] = float((outs["0"].float() - outs["1"].float()).abs().max())
    out[f"extend_{name}"] = res
    print(f"extend {name}: {res}")
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/r02_exp1.json").write_text(json.dumps(out, indent=1))
