"""o_proj / down GEMM + add_rmsnorm combine, per pair, graph-timed over rotating weights: product library vs a
variant (SGLANG_AMD_LIB) with another combine workgroup shape."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def graph_time(fn, reps=20):
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


for name, N, Kd in (("o_proj", 4096, 4096), ("down", 4096, 14336)):
    ws = [torch.randn((N, Kd), device=dev).to(BF) * 0.02 for _ in range(16)]
    x = torch.randn((64, Kd), device=dev).to(BF)
    res = torch.randn((64, N), device=dev).to(BF)
    nw = torch.ones(N, dtype=BF, device=dev)
    t = graph_time(lambda: [K.wstream_gemm(x, w, epilogue="add_rmsnorm", residual=res, norm_weight=nw, eps=1e-5, out_blocked=True) for w in ws]) / 16
    print(name, round(t, 2), "us per GEMM + combine")
