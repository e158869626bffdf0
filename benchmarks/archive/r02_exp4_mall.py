"""Does the 256 MiB Infinity Cache help the decode weight stream?  For each dense projection of the Llama-3-8B
decode layer (M = 64): the wstream GEMM on (a) weights nobody touched since they were evicted, (b) the same
matrix every launch, (c) weights another kernel read just before (a stand-in for a prefetch that runs in the
HBM-idle attention / combine windows) vs (d) the same extra kernel reading some other matrix."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
COPIES = 24


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
x_cache = {}
for name, N, Kd, ep in (("qkv", 6144, 4096, "none"), ("o_proj", 4096, 4096, "none"), ("gate_up", 28672, 4096, "silu_and_mul"),
                        ("down", 4096, 14336, "none")):
    ws = [torch.randn((N, Kd), device=dev).to(BF) * 0.02 for _ in range(COPIES)]
    x = torch.randn((64, Kd), device=dev).to(BF)
    mb = N * Kd * 2 / 1e6
    sink = torch.zeros((COPIES,), dtype=torch.int32, device=dev)

    def gemm(w):
        return K.wstream_gemm(x, w, epilogue=ep)

    def touch(i, w):
        sink[i] = w.view(torch.int32).sum()          # reads every byte of w once

    cold = graph_time(lambda: [gemm(w) for w in ws]) / COPIES
    warm = graph_time(lambda: [gemm(ws[0]) for _ in ws]) / COPIES
    touch_only = graph_time(lambda: [touch(i, w) for i, w in enumerate(ws)]) / COPIES
    pre_same = graph_time(lambda: [(touch(i, w), gemm(w)) for i, w in enumerate(ws)]) / COPIES
    pre_other = graph_time(lambda: [(touch(i, ws[(i + COPIES // 2) % COPIES]), gemm(w)) for i, w in enumerate(ws)]) / COPIES
    out[name] = {"MB": mb, "cold_us": cold, "warm_us": warm, "touch_us": touch_only, "touch_same_then_gemm_us": pre_same,
                 "touch_other_then_gemm_us": pre_other, "cold_TBps": mb / cold, "warm_TBps": mb / warm,
                 "gemm_after_touch_us": pre_same - touch_only}
    print(name, json.dumps(out[name]))
    del ws
    torch.cuda.empty_cache()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp4_mall.json").write_text(json.dumps(out, indent=1))
