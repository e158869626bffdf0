"""Graph-timed extend attention on the bench's cold (4 x 1024, no prefix) and warm (60 x 128 over 896) shapes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
Hq, Hkv, D = 32, 8, 128
slots = 64 * 1200 + 4096
kc = torch.randn((slots, Hkv, D), device=dev).to(BF)
vc = torch.randn((slots, Hkv, D), device=dev).to(BF)
r2t = torch.zeros((65, 1160), dtype=torch.int32, device=dev)
perm = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
for b in range(64):
    r2t[b + 1, :1088] = perm[b * 1088:(b + 1) * 1088]


def graph_time(fn, reps=10):
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


for name, nreq, pre, e in (("cold", 4, 0, 1024), ("warm", 60, 896, 128), ("long", 2, 0, 4096)):
    T = nreq * e
    if pre + e > 1088:
        r2 = torch.zeros((nreq + 1, pre + e), dtype=torch.int32, device=dev)
        pp = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
        for b in range(nreq):
            r2[b + 1] = pp[b * (pre + e):(b + 1) * (pre + e)]
    else:
        r2 = r2t
    qx = torch.randn((T, Hq, D), device=dev).to(BF)
    ox = torch.empty_like(qx)
    seq_x = torch.full((nreq,), pre + e, dtype=torch.int32, device=dev)
    pre_x = torch.full((nreq,), pre, dtype=torch.int32, device=dev)
    qo = (torch.arange(nreq + 1, device=dev) * e).to(torch.int32)
    pool_x = torch.arange(1, nreq + 1, device=dev)
    t = graph_time(lambda: K.extend_attention(qx, ox, kc, vc, r2, pool_x, seq_x, pre_x, qo, e, D ** -0.5, True))
    fl = nreq * 4 * Hq * D * (e * pre + e * (e + 1) / 2)
    print(name, round(t, 1), "us", round(fl / t / 1e6, 1), "TF/s")
