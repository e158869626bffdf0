"""Extend attention workgroup shapes (SGL_AMD_EXTEND_SHAPE = 82 / 42 / 41) on the bench's cold and warm prefill."""
import json, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K
dev = torch.device("cuda:0"); BF = torch.bfloat16
def graph_time(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
Hq, Hkv, D, G, P, prefix = 32, 8, 128, 4, 16, 896
B = G * P; slots = B * 1200 + 4096
kc = torch.randn((slots, Hkv, D), device=dev).to(BF); vc = torch.randn((slots, Hkv, D), device=dev).to(BF)
r2t = torch.zeros((B + 1, 1160), dtype=torch.int32, device=dev)
perm = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
off = 0
for b in range(B):
    r2t[b + 1, :1024] = perm[off: off + 1024]; off += 1024
    r2t[b + 1, :prefix] = r2t[(b // P) * P + 1, :prefix]
out = {}
for name, nreq, pre, e in (("cold", G, 0, 1024), ("warm", B - G, prefix, 128)):
    T = nreq * e
    qx = torch.randn((T, Hq, D), device=dev).to(BF)
    seq_x = torch.full((nreq,), pre + e, dtype=torch.int32, device=dev); pre_x = torch.full((nreq,), pre, dtype=torch.int32, device=dev)
    qo = (torch.arange(nreq + 1, device=dev) * e).to(torch.int32); pool_x = torch.arange(1, nreq + 1, device=dev)
    fl = nreq * 4 * Hq * D * (e * pre + e * (e + 1) / 2)
    for shape in ("82", "42", "41"):
        os.environ["SGL_AMD_EXTEND_SHAPE"] = shape
        ox = torch.empty_like(qx)
        t = graph_time(lambda: K.extend_attention(qx, ox, kc, vc, r2t, pool_x, seq_x, pre_x, qo, e, D ** -0.5, True))
        out[f"{name}_{shape}"] = {"us": t, "tflops": fl / t / 1e6}
        print(name, shape, out[f"{name}_{shape}"])
Path("gpurun_out/r02_exp3.json").write_text(json.dumps(out, indent=1))
