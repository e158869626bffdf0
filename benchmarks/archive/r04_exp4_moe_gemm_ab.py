"""The two grouped GEMMs of fused_experts on their own (graph-timed), 256-row tiles vs 128-row tiles, for whichever
library SGLANG_AMD_LIB points at (benchmarks/build_variant.py: loop form / pinning / ablations)."""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


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


E, topk, Kd = 8, 2, 4096
out = {}
for M, N in ((7680, 7168), (4096, 14336), (7680, 14336)):
    w13 = (torch.randn((E, 2 * N, Kd), device=dev) * 0.03).to(BF)
    w2 = (torch.randn((E, Kd, N), device=dev) * 0.03).to(BF)
    x = torch.randn((M, Kd), device=dev).to(BF)
    tw, ti = K.topk_softmax(torch.randn((M, E), device=dev), topk, True)
    numel = M * topk
    inter = torch.empty((numel, N), dtype=BF, device=dev)
    down = torch.empty((numel, Kd), dtype=torch.float32, device=dev)
    twf = tw.reshape(-1).contiguous()
    row = {}
    for rows in (256, 128):
        s, e, post = K.moe_align_block_size(ti, 256, E)
        t_up = graph_time(lambda: K.moe_tiled_gemm(x, w13, inter, s, e, post, None, False, topk, numel, 256, fuse_silu=True, tile_rows=rows))
        t_dn = graph_time(lambda: K.moe_tiled_gemm(inter, w2, down, s, e, post, twf, True, 1, numel, 256, round_before_scale=True, tile_rows=rows))
        row[f"up{rows}"] = round(numel * 2 * N * Kd * 2 / t_up / 1e12, 1)
        row[f"down{rows}"] = round(numel * N * Kd * 2 / t_dn / 1e12, 1)
        row[f"up{rows}_us"], row[f"down{rows}_us"] = round(t_up * 1e6, 1), round(t_dn * 1e6, 1)
    out[f"M{M}_N{N}"] = row
    print(os.environ.get("SGLANG_AMD_LIB", "default").split("lib_")[-1], M, N, row)
    del w13, w2
Path("gpurun_out").mkdir(exist_ok=True)
tag = os.environ.get("SGLANG_AMD_LIB", "default").split("lib_")[-1].replace(".so", "")
Path(f"gpurun_out/r04_exp4_moe_gemm_ab_{tag}.json").write_text(json.dumps(out, indent=1))
