"""(waves per group, K splits) of the three fused decode projections in their real form (GEMM + the combine that
carries rope / add+norm), hipGraph-timed over rotating weights, M = 64, Llama-3-8B shapes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

BF = torch.bfloat16
DEV = torch.device("cuda:0")


def timeit(fn, iters=16, reps=5):
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
    M, H, I, D, Hq, Hkv = 64, 4096, 14336, 128, 32, 8
    copies = 12
    res = torch.randn((M, H), device=DEV).to(BF)
    nw_ = torch.ones(H, device=DEV, dtype=BF)
    it = [0]

    def rot(ws):
        it[0] += 1
        return ws[it[0] % copies]

    cands = [(4, 2), (4, 3), (4, 4), (5, 3), (5, 4), (6, 3), (6, 4), (7, 4), (8, 4), (8, 6), (8, 8), (4, 6), (4, 8)]
    for name, N, Kd in (("o_proj+norm", H, H), ("down+norm", H, I)):
        ws = [(torch.randn((N, Kd), device=DEV) * 0.02).to(BF) for _ in range(copies)]
        x = torch.randn((M, Kd), device=DEV).to(BF)
        out = []
        for nw, s in cands:
            if s > Kd // 256:
                continue
            t = timeit(lambda: K.wstream_gemm(x, rot(ws), epilogue="add_rmsnorm", residual=res, norm_weight=nw_, eps=1e-5,
                                              waves_per_group=nw, splits=s))
            out.append((t, nw, s))
        out.sort()
        print(name, "auto", K.choose_wstream_config(M, N, Kd, True), [(nw, s, round(t, 1)) for t, nw, s in out[:6]], flush=True)
        del ws
    N = (Hq + 2 * Hkv) * D
    ws = [(torch.randn((N, H), device=DEV) * 0.02).to(BF) for _ in range(copies)]
    x = torch.randn((M, H), device=DEV).to(BF)
    pos = torch.randint(0, 2048, (M,), device=DEV)
    cache = torch.randn((4096, D), device=DEV).to(BF)
    kc = torch.zeros((M + 8, Hkv, D), dtype=BF, device=DEV)
    vc = torch.zeros_like(kc)
    loc = torch.arange(1, M + 1, device=DEV)
    out = []
    for nw, s in cands:
        t = timeit(lambda: K.wstream_qkv_rope(x, rot(ws), None, pos, cache, Hq, Hkv, D, kc, vc, loc, waves_per_group=nw, splits=s))
        out.append((t, nw, s))
    out.sort()
    print("qkv+rope", "auto", K.choose_wstream_config(M, N, H, True), [(nw, s, round(t, 1)) for t, nw, s in out[:6]], flush=True)


if __name__ == "__main__":
    main()
