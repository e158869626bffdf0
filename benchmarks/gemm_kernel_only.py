"""Launch only the dominant hand-written kernel (the weight-streaming gate_up GEMM with the silu_and_mul
epilogue) at bench.py's roofline shape, for `rocprofv3 --pmc ...` passes (HBM traffic per launch).
The weights rotate over 4 copies of 235 MB so that every launch streams from HBM, not the infinity cache."""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=12)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, N, Kd = 64, 28672, 4096
    ws = [(torch.randn((N, Kd), device=dev) * 0.02).to(torch.bfloat16) for _ in range(4)]
    x = torch.randn((M, Kd), device=dev).to(torch.bfloat16)
    for i in range(a.iters):
        K.wstream_gemm(x, ws[i % 4], epilogue="silu_and_mul")
    torch.cuda.synchronize()
    print(f"algorithmic bytes/launch: weights {N * Kd * 2} + x {M * Kd * 2} + y {M * N}")


if __name__ == "__main__":
    main()
