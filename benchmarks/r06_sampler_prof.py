"""Round 6: the one-call sampler at [64, 128256] under rocprofv3 --kernel-trace (per-launch durations of its four kernels)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
B, V = 64, 128256
lg = (torch.randn((B, V), device=dev) * 2.0).to(torch.bfloat16)
t = torch.ones((B, 1), device=dev)
tk = torch.full((B,), 50, dtype=torch.int32, device=dev)
tp = torch.full((B,), 0.9, device=dev)
seed = torch.arange(B, device=dev, dtype=torch.int64) + 1234
pos = torch.full((B,), 1024, dtype=torch.int64, device=dev)
for _ in range(20):
    ids, fb = K.sample_from_bf16_logits(lg, t, tk, tp, None, seed, pos, return_fallback=True)
torch.cuda.synchronize()
print("rows redone the long way:", int(fb.sum()))

