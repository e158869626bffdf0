"""Extend attention: the 32x32 two-score-set kernel (default) against the ping-pong kernel (debug flag 2) on the bench's
prefill shapes."""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K, native  # noqa: E402

BF, DEV = torch.bfloat16, torch.device("cuda:0")


def graph_time(fn, reps=10, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * iters) * 1e-3


def main():
    Hq, Hkv, D = 32, 8, 128
    out = {}
    for name, nreq, pre, e in (("cold", 4, 0, 1024), ("warm", 60, 896, 128), ("long", 2, 0, 4096), ("llava", 8, 0, 2952)):
        ctx = pre + e
        slots = nreq * ctx + 8
        kc = torch.randn((slots, Hkv, D), device=DEV).to(BF)
        vc = torch.randn((slots, Hkv, D), device=DEV).to(BF)
        r2t = torch.zeros((nreq + 1, ctx + 8), dtype=torch.int32, device=DEV)
        perm = (torch.randperm(slots - 1, device=DEV) + 1).to(torch.int32)
        for b in range(nreq):
            r2t[b + 1, :ctx] = perm[b * ctx:(b + 1) * ctx]
        T = nreq * e
        q = torch.randn((T, Hq, D), device=DEV).to(BF)
        seq = torch.full((nreq,), ctx, dtype=torch.int32, device=DEV)
        prefix = torch.full((nreq,), pre, dtype=torch.int32, device=DEV)
        qo = (torch.arange(nreq + 1, device=DEV) * e).to(torch.int32)
        pool = torch.arange(1, nreq + 1, device=DEV)
        fl = nreq * 4 * Hq * D * (e * pre + e * (e + 1) / 2)
        rec = {}
        outs = {}
        for label, shape, flags in (("pingpong", 0, 2), ("form32", 0, 0)):
            native.call("sgl_amd_debug_extend_attention_shape", shape, flags)
            o = torch.empty_like(q)
            t = graph_time(lambda: K.extend_attention(q, o, kc, vc, r2t, pool, seq, prefix, qo, e, D ** -0.5, True))
            outs[label] = o.float()
            rec[label] = {"us": t * 1e6, "tflops": fl / t / 1e12, "frac": fl / t / 1e12 / 2500.0}
        rec["max_abs_diff"] = max(float((outs["pingpong"] - outs[k]).abs().max()) for k in ("form32",))
        native.call("sgl_amd_debug_extend_attention_shape", 0, 0)
        out[name] = rec
        print(name, {k: (round(v['us'], 1), round(v['frac'], 3)) if isinstance(v, dict) else v for k, v in rec.items()})
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/r03_exp4_extend_32x32%s.json" % os.environ.get("EXP_TAG", "")).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
