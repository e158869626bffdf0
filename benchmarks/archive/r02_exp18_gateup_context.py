"""Why does bench.py's graph-timed gate_up launch take 43-44 us when the same launch inside the decode step takes 39 us
(rocprof)?  Same call on 32 fresh weight copies: back-to-back, with the down projection between launches, and with a
small kernel between launches."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
L, M, H, I = 32, 64, 4096, 14336


def graph_time(fn, reps=5):
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


wg = [torch.randn((2 * I, H), device=dev).to(BF) * 0.02 for _ in range(L)]
wd = [torch.randn((H, I), device=dev).to(BF) * 0.02 for _ in range(L)]
xr = torch.randn((M, H), device=dev).to(BF)
xb = K.blocked_activation(M, H, dev)
xb.copy_(torch.randn(xb.shape, device=dev).to(BF))
res = torch.randn((M, H), device=dev).to(BF)
nw = torch.ones(H, device=dev).to(BF)
small = torch.zeros(1024, device=dev)
out = {}
out["row_major_back_to_back"] = graph_time(lambda: [K.wstream_gemm(xr, w, epilogue="silu_and_mul") for w in wg]) / L
out["blocked_back_to_back"] = graph_time(lambda: [K.wstream_gemm(xb, w, epilogue="silu_and_mul", out_blocked=True) for w in wg]) / L
t_small = graph_time(lambda: [small.add_(1.0) for _ in wg]) / L
out["small_kernel"] = t_small
out["blocked_with_small_between"] = graph_time(lambda: [(K.wstream_gemm(xb, w, epilogue="silu_and_mul", out_blocked=True), small.add_(1.0)) for w in wg]) / L - t_small


def layer(w1, w2):
    a = K.wstream_gemm(xb, w1, epilogue="silu_and_mul", out_blocked=True)
    return K.wstream_gemm(a, w2, epilogue="add_rmsnorm", residual=res, norm_weight=nw, eps=1e-5, out_blocked=True)


t_down = graph_time(lambda: [K.wstream_gemm(K.blocked_activation(M, I, dev), w2, epilogue="add_rmsnorm", residual=res, norm_weight=nw,
                                            eps=1e-5, out_blocked=True) for w2 in wd]) / L
out["down_pair_alone"] = t_down
out["gate_up_plus_down_pair"] = graph_time(lambda: [layer(a, b) for a, b in zip(wg, wd)]) / L
for k, v in out.items():
    print(k, round(v, 2))
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp18_gateup_context.json").write_text(json.dumps(out, indent=1))
