"""Is the shared activation image an L2-channel hot spot?  x [64, K] row-major puts a K chunk's 64 x 256 B pieces 2K
bytes apart (one or two L2 channels for all 256 CUs); chunk-major packing [K/128][64][128] makes a chunk 16 KiB
contiguous."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from sglang_amd import kernels as K, native  # noqa: E402

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
for name, N, Kd, ep in (("qkv", 6144, 4096, 0), ("o_proj", 4096, 4096, 0), ("gate_up", 28672, 4096, 1), ("down", 4096, 14336, 0)):
    M = 64
    ws = [torch.randn((N, Kd), device=dev).to(BF) * 0.02 for _ in range(COPIES)]
    x = torch.randn((M, Kd), device=dev).to(BF)
    xp = x.view(M, Kd // 128, 128).permute(1, 0, 2).contiguous()
    n_out = N // 2 if ep else N
    y0 = torch.empty((M, n_out), dtype=BF, device=dev)
    y1 = torch.empty_like(y0)
    nw, s = K.choose_wstream_config(M, N, Kd, False, ep == 1)
    wsp = torch.empty((s * M * N,), dtype=torch.float32, device=dev)

    epi = "silu_and_mul" if ep else "none"
    t_row = graph_time(lambda: [K.wstream_gemm(x, w, epilogue=epi, out=y0) for w in ws]) / COPIES
    t_pack = graph_time(lambda: [K.wstream_gemm(xp, w, epilogue=epi, out=y1) for w in ws]) / COPIES
    out[name] = {"MB": N * Kd * 2 / 1e6, "nw": nw, "splits": s, "row_major_us": t_row, "chunk_major_us": t_pack,
                 "same": bool(torch.equal(y0, y1)), "row_TBps": N * Kd * 2 / 1e6 / t_row, "chunk_TBps": N * Kd * 2 / 1e6 / t_pack}
    print(name, json.dumps(out[name]))
    del ws
    torch.cuda.empty_cache()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp6_xpack.json").write_text(json.dumps(out, indent=1))
