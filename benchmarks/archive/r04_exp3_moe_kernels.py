"""Per-kernel times of fused_experts at a prefill batch (run under rocprofv3 --kernel-trace --stats via benchmarks/prof_cmd.sh)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
E, topk, N, Kd = 8, 2, int(sys.argv[2]) if len(sys.argv) > 2 else 7168, 4096
M = int(sys.argv[1]) if len(sys.argv) > 1 else 7680
w13 = (torch.randn((E, 2 * N, Kd), device=dev) * 0.03).to(BF)
w2 = (torch.randn((E, Kd, N), device=dev) * 0.03).to(BF)
x = torch.randn((M, Kd), device=dev).to(BF)
tw, ti = K.topk_softmax(torch.randn((M, E), device=dev), topk, True)
for _ in range(12):
    K.fused_experts(x, w13, w2, tw, ti)
torch.cuda.synchronize()
print("flops per call", M * topk * 3 * N * Kd * 2, "up", M * topk * 2 * N * Kd * 2, "down", M * topk * N * Kd * 2)
