"""Mixtral-8x7B expert GEMMs at decode batch sizes: weight-streaming grouped form vs the skinny kernel (hipGraph-timed)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

BF = torch.bfloat16
DEV = torch.device("cuda:0")


def timeit(fn, iters=8, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * reps) * 1e3


def main():
    E, k, N, Kd = 8, 2, 14336, 4096
    w13 = (torch.randn((E, 2 * N, Kd), device=DEV) * 0.02).to(BF)
    w2 = (torch.randn((E, Kd, N), device=DEV) * 0.02).to(BF)
    for M in (16, 64):
        x = torch.randn((M, Kd), device=DEV).to(BF)
        ids = torch.argsort(torch.rand((M, E), device=DEV), dim=1)[:, :k].to(torch.int32)
        tw = torch.rand((M, k), device=DEV)
        bm = K.choose_moe_block_m(M * k, E)
        s, e, post = K.moe_align_block_size(ids, bm, E)
        inter = torch.empty((M * k, N), dtype=BF, device=DEV)
        down = torch.empty((M * k, Kd), dtype=torch.float32, device=DEV)
        hit = len(set(ids.flatten().tolist()))
        for name, fn in (("wstream", K.moe_wstream_gemm), ("skinny ", K.moe_grouped_gemm)):
            t_up = timeit(lambda: fn(x, w13, inter, s, e, post, None, False, k, M * k, bm, fuse_silu=True))
            t_dn = timeit(lambda: fn(inter, w2, down, s, e, post, tw.reshape(-1), True, 1, M * k, bm, round_before_scale=True))
            up_b, dn_b = hit * 2 * N * Kd * 2, hit * N * Kd * 2
            print(f"M={M:3d} block_m={bm} experts hit={hit} {name}: up {t_up:7.1f} us ({up_b / t_up / 1e3:5.0f} GB/s)  "
                  f"down {t_dn:7.1f} us ({dn_b / t_dn / 1e3:5.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
