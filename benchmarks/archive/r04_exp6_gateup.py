"""gate_up + silu_and_mul at the decode batch: the two-tile form (4 waves x (gate tile + up tile) per workgroup: 224 workgroups
on Llama-3-8B) against the one-tile interleaved form (7 waves x one 8 + 8-row tile: 256 workgroups).  Each launch streams a
different weight matrix (32 of them, like the model's layers), chunk-major activations, captured in one graph."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def graph_time(fn, launches, reps=10, warm=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(warm):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * launches) * 1e3


out = {}
for name, M, N, Kd, copies in (("llama-3-8b", 64, 28672, 4096, 32), ("llama-3-8b M=16", 16, 28672, 4096, 32), ("llama-3-70b", 64, 57344, 8192, 12)):
    ws = [(torch.randn((N, Kd), device=dev) * 0.02).to(BF) for _ in range(copies)]
    x = K.blocked_activation(M, Kd, dev)
    x.copy_(torch.randn(x.shape, device=dev).to(BF))
    row = {}
    for label, nw, tpw in (("two_tile_4w", 4, 2), ("interleaved_7w", 7, 1), ("interleaved_8w", 8, 1), ("interleaved_6w", 6, 1)):
        t = graph_time(lambda: [K.wstream_gemm(x, w, epilogue="silu_and_mul", splits=1, waves_per_group=nw, tiles_per_wave=tpw, out_blocked=True) for w in ws], copies)
        alg = N * Kd * 2 + M * Kd * 2 + M * (N // 2) * 2
        row[label] = {"us": round(t, 2), "TB/s": round(alg / t / 1e6, 3)}
    row["auto"] = list(K.choose_wstream_decomposition(M, N, Kd, True, True))
    out[name] = row
    print(name, row)
    del ws
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/r04_exp6_gateup_interleaved.json").write_text(json.dumps(out, indent=1))
