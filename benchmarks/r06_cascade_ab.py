"""Round 6: the shared-prefix chunk kernel + merge on the bench batch, 8 layers' pools rotated (rows come from HBM); request tables of
1280 tokens (one workgroup per unit) and 8192 tokens (the looping form the reference's scheduler gets)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K, native  # noqa: E402
from benchmarks.cascade_micro import DEV, BF, slot_table, timeit  # noqa: E402

B, P, Hq, Hkv, D, prefix, unique, ctx, L = 64, 16, 32, 8, 128, 896, 128, 1088, 8
out = {}
for width in (1280, 8192):
    r2t, slots = slot_table("allocator", B, P, ctx, prefix, unique, width)
    kcs = [torch.randn((slots, Hkv, D), device=DEV).to(BF) for _ in range(L)]
    vcs = [torch.randn((slots, Hkv, D), device=DEV).to(BF) for _ in range(L)]
    q = torch.randn((B, Hq, D), device=DEV).to(BF)
    pool = torch.arange(1, B + 1, device=DEV)
    seq = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
    ws = K.CascadeWorkspace(B, Hq, D, width, DEV)
    K.cascade_plan(ws, r2t, pool, seq, Hq, Hkv)
    res = {}
    for rep in range(3):
        o = torch.empty_like(q)
        state = {"i": 0}

        def call():
            i = state["i"] = (state["i"] + 1) % L
            K.cascade_decode_attention(ws, q, kcs[i], vcs[i], o, r2t, pool, seq, D ** -0.5)

        t = timeit(call, iters=16)
        res.setdefault("us", []).append(round(t, 2))
        print(f"width {width}: {t:.2f} us", flush=True)
    # against the plain paged decode kernel (no prefix sharing): same attention, another reduction structure
    o2 = torch.empty_like(q)
    K.cascade_decode_attention(ws, q, kcs[0], vcs[0], o, r2t, pool, seq, D ** -0.5)
    K.decode_attention(q, kcs[0], vcs[0], o2, r2t, pool, seq, D ** -0.5, 1, None, None)
    torch.cuda.synchronize()
    res["max_abs_diff_vs_plain_decode_kernel"] = float((o.float() - o2.float()).abs().max())
    print("max |cascade - plain|", res["max_abs_diff_vs_plain_decode_kernel"], flush=True)
    out[str(width)] = res
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps({"what": "cascade chunk + merge us per layer, bench batch, by request-table width", "by_width": out}, indent=1))
