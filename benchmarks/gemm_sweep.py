"""Sweep of the weight-streaming decode GEMM's decomposition on one MI355X vs hipBLASLt (F.linear).

    python benchmarks/gemm_sweep.py [--json gpurun_out/gemm_sweep.json] [--quick]
"""
import argparse
import json
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

BF = torch.bfloat16
DEV = torch.device("cuda:0")


def timeit(fn, iters=24, warmup=3, reps=5):
    """Kernel time without host launch overhead: `iters` calls captured in one hipGraph, replayed."""
    for _ in range(warmup):
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
    return s.elapsed_time(e) / (iters * reps) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--M", type=int, default=64)
    ap.add_argument("--blocked", action="store_true", help="chunk-major [K/128, M, 128] activations")
    args = ap.parse_args()
    shapes = [("gate_up", 28672, 4096), ("down", 4096, 14336), ("qkv", 6144, 4096), ("o", 4096, 4096),
              ("lm_head", 128256, 4096)]
    M = args.M
    rows = []
    # rotate over several weight copies so the stream comes from HBM, not the 256 MiB infinity cache
    for name, N, Kd in shapes:
        copies = max(2, int(600e6 // (N * Kd * 2)) + 1)
        ws = [(torch.randn((N, Kd), device=DEV) * 0.02).to(BF) for _ in range(copies)]
        x = torch.randn((M, Kd), device=DEV).to(BF)
        if args.blocked:
            x = x.view(M, Kd // 128, 128).permute(1, 0, 2).contiguous()
        wbytes = N * Kd * 2
        it = [0]

        def nxt():
            it[0] = (it[0] + 1) % copies
            return ws[it[0]]

        t = timeit(lambda: F.linear(K.unblock(x), nxt()))
        rows.append(dict(shape=name, N=N, K=Kd, M=M, impl="hipblaslt", us=t * 1e6, GBps=wbytes / t / 1e9))
        print(json.dumps(rows[-1]), flush=True)
        nch = Kd // 128
        cands = set()
        auto = K.choose_wstream_config(M, N, Kd)
        cands.add(auto)
        if not args.quick:
            for nw in (4, 5, 6, 7, 8):
                for s in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16):
                    if s <= nch // 2:
                        cands.add((nw, s))
            for nw in (2, 3, 4):                      # two tiles per wave
                for s in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16):
                    if s <= nch // 2 and N % 32 == 0:
                        cands.add((nw, s, 2))
        for cand in sorted(cands):
            nw, s = cand[:2]
            tpw = cand[2] if len(cand) > 2 else 1
            try:
                t = timeit(lambda: K.wstream_gemm(x, nxt(), waves_per_group=nw, splits=s, tiles_per_wave=tpw))
            except RuntimeError as e:
                print("skip", nw, s, e)
                continue
            rows.append(dict(shape=name, N=N, K=Kd, M=M, impl="wstream", nw=nw, splits=s, tpw=tpw, auto=(nw, s) == auto and tpw == 1,
                             us=t * 1e6, GBps=wbytes / t / 1e9))
            print(json.dumps(rows[-1]), flush=True)
        del ws
        torch.cuda.empty_cache()
    if args.json:
        Path(args.json).parent.mkdir(parents=True, exist_ok=True)
        Path(args.json).write_text(json.dumps(rows, indent=1))
    # best per shape
    for name, _, _ in shapes:
        rs = [r for r in rows if r["shape"] == name]
        lib = [r for r in rs if r["impl"] == "hipblaslt"][0]
        best = min((r for r in rs if r["impl"] == "wstream"), key=lambda r: r["us"])
        print(f"{name:8s} hipblaslt {lib['us']:7.1f} us {lib['GBps']:6.0f} GB/s | best wstream nw={best['nw']} tpw={best['tpw']} s={best['splits']} "
              f"{best['us']:7.1f} us {best['GBps']:6.0f} GB/s")


if __name__ == "__main__":
    main()
