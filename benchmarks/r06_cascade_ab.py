"""Round 6: write-through partial stores (flags bit 0) and staggered row bursts (flags bits 8.., 10 ns ticks per dispatch round) of the
shared-prefix chunk kernel, A/B on the bench batch (8 layers' pools rotated: rows come from HBM)."""
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
    for flags in (0, 1, 0, 1) + tuple(1 | (t << 8) for t in (50, 100, 150, 200, 250, 300, 400)) + tuple(t << 8 for t in (150, 250)):
        native.lib().sgl_amd_debug_cascade_flags(flags)
        o = torch.empty_like(q)
        state = {"i": 0}

        def call():
            i = state["i"] = (state["i"] + 1) % L
            K.cascade_decode_attention(ws, q, kcs[i], vcs[i], o, r2t, pool, seq, D ** -0.5)

        t = timeit(call, iters=16)
        K.cascade_decode_attention(ws, q, kcs[0], vcs[0], o, r2t, pool, seq, D ** -0.5)
        torch.cuda.synchronize()
        res.setdefault(flags, []).append((t, o.clone()))
        print(f"width {width} wt {flags & 1} stagger {(flags >> 8) / 100:.1f} us/round: {t:.2f} us", flush=True)
    for f, v in res.items():
        assert torch.equal(res[0][0][1], v[0][1]), f"flags {f} changed the result"
    out[str(width)] = {str(f): [round(t, 2) for t, _ in v] for f, v in res.items()}
native.lib().sgl_amd_debug_cascade_flags(0)
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps({"what": "cascade chunk + merge us per layer, bench batch; key = flags: bit 0 write-through partials, bits 8.. = stagger in 10 ns ticks per dispatch round", "us": out}, indent=1))
