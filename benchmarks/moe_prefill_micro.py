"""Mixtral-8x7B TP=2 rank expert GEMMs at a prefill batch: the row-tiled MFMA grouped GEMM (moe_tiled_gemm.hip)
against the weight-streaming form it replaces for large M.  hipGraph replays between events."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
E, topk, N, Kd = 8, 2, 7168, 4096


def graph_time(fn, reps=5):
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
    return e0.elapsed_time(e1) / reps * 1e-3


w13 = (torch.randn((E, 2 * N, Kd), device=dev) * 0.03).to(BF)
w2 = (torch.randn((E, Kd, N), device=dev) * 0.03).to(BF)
res = {}
for M in (2048, 4096, 7680, 16384):
    x = torch.randn((M, Kd), device=dev).to(BF)
    logits = torch.randn((M, E), device=dev)
    tw, ti = K.topk_softmax(logits, topk, True)
    flops = M * topk * 3 * N * Kd * 2
    t_tiled = graph_time(lambda: K.fused_experts(x, w13, w2, tw, ti))
    plans = {}
    for plan in ((128, 128, 128), (256, 256, 128), (256, 256, 256), (256, 128, 256)):
        K.MOE_TILE_OVERRIDE = plan
        t = graph_time(lambda: K.fused_experts(x, w13, w2, tw, ti))
        plans["align%d_up%d_down%d" % plan] = {"ms": t * 1e3, "tflops": flops / t / 1e12}
    K.MOE_TILE_OVERRIDE = None
    old = K.MOE_TILED_MIN_ROWS_PER_EXPERT
    K.MOE_TILED_MIN_ROWS_PER_EXPERT = 1 << 30
    t_stream = graph_time(lambda: K.fused_experts(x, w13, w2, tw, ti), reps=2)
    K.MOE_TILED_MIN_ROWS_PER_EXPERT = old
    res[M] = {"tiled_ms": t_tiled * 1e3, "tiled_tflops": flops / t_tiled / 1e12, "plan": K.moe_tile_plan(M * topk, E, N, Kd),
              "plans": plans, "wstream_ms": t_stream * 1e3, "wstream_tflops": flops / t_stream / 1e12}
    print(M, res[M])
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/r04_moe_prefill_micro.json").write_text(json.dumps(res, indent=1))
