"""Per-CU or chip-level limit?  The same weight stream per workgroup on 128 / 192 / 224 / 256 workgroups (one per CU)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


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
for M in (64, 16):
    for Kd in (4096, 16384):
        for wgs in (64, 128, 192, 224, 256, 512):
            N = wgs * 4 * 16
            copies = max(2, int(1.5e9 // (N * Kd * 2)))
            ws = [torch.randn((N, Kd), device=dev).to(BF) * 0.02 for _ in range(copies)]
            x = torch.randn((M, Kd), device=dev).to(BF)
            t = graph_time(lambda: [K.wstream_gemm(x, w, waves_per_group=4, splits=1) for w in ws]) / copies
            mb = N * Kd * 2 / 1e6
            out[f"M{M}_K{Kd}_wg{wgs}"] = {"MB": mb, "us": t, "TBps": mb / t, "GBps_per_wg": mb / t / wgs * 1e3}
            print(f"M{M}_K{Kd}_wg{wgs}", json.dumps(out[f"M{M}_K{Kd}_wg{wgs}"]))
            del ws
            torch.cuda.empty_cache()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp7_cu_scaling.json").write_text(json.dumps(out, indent=1))
