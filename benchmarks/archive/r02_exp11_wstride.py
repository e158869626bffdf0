"""Does the weight row stride matter?  Split-K projections with rows K elements apart (a multiple of 4 KiB) vs padded
rows (K + 128 / K + 64 elements), same bytes streamed."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
COPIES = 20


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


out = {}
for name, N, Kd in (("qkv", 6144, 4096), ("o_proj", 4096, 4096), ("down", 4096, 14336), ("gate_up", 28672, 4096)):
    x = torch.randn((64, Kd), device=dev).to(BF)
    xb = x.view(64, Kd // 128, 128).permute(1, 0, 2).contiguous()
    ep = "silu_and_mul" if name == "gate_up" else "none"
    row = {}
    for pad in (0, 64, 128, 192):
        ws = [torch.randn((N, Kd + pad), device=dev).to(BF)[:, :Kd] * 0.02 for _ in range(COPIES)]
        row[f"pad{pad}"] = graph_time(lambda: [K.wstream_gemm(xb, w, epilogue=ep) for w in ws]) / COPIES
        del ws
        torch.cuda.empty_cache()
    out[name] = row
    print(name, json.dumps({k: round(v, 2) for k, v in row.items()}))
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp11_wstride.json").write_text(json.dumps(out, indent=1))
