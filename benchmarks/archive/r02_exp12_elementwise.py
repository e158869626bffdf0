"""Prefill-sized elementwise kernels: silu_and_mul [T, 28672] -> [T, 14336], fused_add_rmsnorm [T, 4096]."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
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


for T in (4096, 7680):
    xs = [torch.randn((T, 28672), device=dev).to(BF) for _ in range(3)]
    out = torch.empty((T, 14336), dtype=BF, device=dev)
    t = graph_time(lambda: [K.silu_and_mul(x, out) for x in xs]) / 3
    print(f"silu_and_mul T={T}: {t:.1f} us, {T * 14336 * 6 / t / 1e6:.2f} TB/s")
    hs = [torch.randn((T, 4096), device=dev).to(BF) for _ in range(6)]
    res = torch.randn((T, 4096), device=dev).to(BF)
    w = torch.ones(4096, dtype=BF, device=dev)
    t = graph_time(lambda: [K.fused_add_rmsnorm(h, res, w, 1e-5) for h in hs]) / 6
    print(f"fused_add_rmsnorm T={T}: {t:.1f} us, {T * 4096 * 8 / t / 1e6:.2f} TB/s")

from sglang_amd.layers.rotary_embedding import get_rope  # noqa: E402

for T in (4096, 7680):
    Hq, Hk, D = 32, 8, 128
    q = torch.randn((T, Hq * D), device=dev).to(BF)
    k = torch.randn((T, Hk * D), device=dev).to(BF)
    v = torch.randn((T, Hk * D), device=dev).to(BF)
    kc = torch.zeros((T + 16, Hk, D), dtype=BF, device=dev)
    vc = torch.zeros_like(kc)
    loc = torch.randperm(T, device=dev) + 1
    pos = torch.arange(T, device=dev)
    rope = get_rope(D, D, 8192, 500000.0, True, None, BF, dev)
    t = graph_time(lambda: K.rotary_embedding(pos, q, k, D, rope.cos_sin_cache, True, value=v, k_cache=kc, v_cache=vc, cache_loc=loc))
    mb = T * (Hq * D * 4 + Hk * D * 4 + Hk * D * 2 + Hk * D * 4) / 1e6
    print(f"rope + kv store T={T}: {t:.1f} us, {mb / t:.2f} TB/s")
