"""Launch only the dominant hand-written kernel (paged decode attention) at bench.py's roofline shape,
for `rocprofv3 --pmc ...` passes (HBM traffic per launch).  Same slot pattern as bench.py: shared-prefix
rows point at the group leader's slots."""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402
from sglang_amd.layers.attention.hip_backend import choose_num_splits  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-shared-prefix", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, P, Hq, Hkv, D, prefix, len_k = 64, 16, 32, 8, 128, 896, 1088
    slots = B * len_k + 4096
    kc = torch.randn((slots, Hkv, D), device=dev).to(torch.bfloat16)
    vc = torch.randn((slots, Hkv, D), device=dev).to(torch.bfloat16)
    r2t = torch.zeros((B + 1, len_k + 72), dtype=torch.int32, device=dev)
    perm = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
    off = 0
    for b in range(B):
        r2t[b + 1, :len_k] = perm[off: off + len_k]
        off += len_k
        if not a.no_shared_prefix:
            leader = (b // P) * P
            r2t[b + 1, :prefix] = r2t[leader + 1, :prefix]
    pool = torch.arange(1, B + 1, device=dev)
    seq = torch.full((B,), len_k, dtype=torch.int32, device=dev)
    q = torch.randn((B, Hq, D), device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    splits = choose_num_splits(B, Hkv, Hq // Hkv, len_k)
    ws = K.decode_workspace(B, Hq, D, splits, dev) if splits > 1 else (None, None)
    for _ in range(a.iters):
        K.decode_attention(q, kc, vc, o, r2t, pool, seq, D ** -0.5, splits, ws[0], ws[1])
    torch.cuda.synchronize()
    unique = (4 * prefix + B * (len_k - prefix)) if not a.no_shared_prefix else B * len_k
    print(f"algorithmic bytes/launch (no dedup) {B * len_k * 2 * Hkv * D * 2}, unique rows bytes {unique * 2 * Hkv * D * 2}, splits {splits}")


if __name__ == "__main__":
    main()
