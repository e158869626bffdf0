"""The once-per-step shared-prefix plan on the bench batch, graph-timed."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
Hq, Hkv, D, G, P, prefix, len_k, ctx = 32, 8, 128, 4, 16, 896, 1088, 1160
for G in (4, 8):
    B = G * P
    slots = B * 1200 + 4096
    r2t = torch.zeros((B + 1, ctx), dtype=torch.int32, device=dev)
    perm = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
    off = 0
    for b in range(B):
        r2t[b + 1, :len_k] = perm[off: off + len_k]
        off += len_k
        r2t[b + 1, :prefix] = r2t[(b // P) * P + 1, :prefix]
    pool = torch.arange(1, B + 1, device=dev)
    seq = torch.full((B,), len_k, dtype=torch.int32, device=dev)
    ws = K.CascadeWorkspace(B, Hq, D, ctx, dev)

    def fn():
        for _ in range(10):
            K.cascade_plan(ws, r2t, pool, seq, Hq, Hkv)

    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B}: plan {e0.elapsed_time(e1) / 100 * 1e3:.1f} us, items {int(ws.plan[0])}")
