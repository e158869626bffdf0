"""Round 5, experiment 1: the shared-prefix chunk kernel against the WIDTH of the request table (its worst-case grid).

The bench's harness sizes req_to_token for the job (1.2 k tokens); the reference's scheduler sizes it for the model's context
length (8 k for Llama-3-8B, 128 k for Llama-3.1).  Same batch (4 x 16 requests, 896 shared + 192 own tokens), eight layers' K/V
rotated so that the rows come from HBM, hipGraph-timed; forms: one workgroup per unit (grid = worst case of the table) vs a
looping grid of 1024 / 1280 / 2048 workgroups.  Output: gpurun_out/r05_exp1_cascade_table_width.json
"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from benchmarks.cascade_micro import slot_table  # noqa: E402
from sglang_amd import kernels as K, native  # noqa: E402

BF = torch.bfloat16
DEV = torch.device("cuda:0")
L = 8


def timeit(fn, iters=L * 4, reps=5):
    for i in range(L):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i)
    for _ in range(3):
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
    B, P, Hq, Hkv, D, prefix, unique, ctx = 64, 16, 32, 8, 128, 896, 128, 1088
    rows = []
    kcs = vcs = None
    for width in (1160, 8192, 131072):
        r2t, slots = slot_table("allocator", B, P, ctx, prefix, unique, width)
        if kcs is None:
            kcs = [torch.randn((slots, Hkv, D), device=DEV).to(BF) for _ in range(L)]
            vcs = [torch.randn((slots, Hkv, D), device=DEV).to(BF) for _ in range(L)]
        pool = torch.arange(1, B + 1, device=DEV)
        seq = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
        q = torch.randn((B, Hq, D), device=DEV).to(BF)
        outs = {}
        ws = K.CascadeWorkspace(B, Hq, D, width, DEV)
        t_plan = timeit(lambda i: K.cascade_plan(ws, r2t, pool, seq, Hq, Hkv))
        uniq = (4 * prefix + B * (ctx - prefix)) * 2 * Hkv * D * 2
        for form, (single, grid) in (("single-shot", (1 << 40, 0)), ("loop-1024", (1, 1024)), ("loop-1280", (1, 1280)), ("loop-2048", (1, 2048)),
                                     ("default", (0, 0))):    # (round 5 also ran a five-workgroups-per-CU looping instance: profiles/r05_exp1b_cascade_forms.json)
            if form == "single-shot" and width > 32768:
                continue                               # ~1 M workgroups per launch: known, not worth the GPU seconds
            native.call("sgl_amd_debug_cascade_launch_form", single, grid)
            out = torch.empty_like(q)
            t = timeit(lambda i: K.cascade_decode_attention(ws, q, kcs[i % L], vcs[i % L], out, r2t, pool, seq, D ** -0.5))
            K.cascade_decode_attention(ws, q, kcs[0], vcs[0], out, r2t, pool, seq, D ** -0.5)
            torch.cuda.synchronize()
            outs[form] = out.clone()
            rows.append(dict(table_width=width, form=form, chunk_plus_merge_us=round(t, 2), plan_us=round(t_plan, 2),
                             unique_GBps=round(uniq / t / 1e3), max_items=ws.max_items))
            print(rows[-1], flush=True)
        native.call("sgl_amd_debug_cascade_launch_form", 0, 0)
        ref = outs["default"]
        for form, o in outs.items():
            assert torch.equal(o, ref), f"{form} differs from the default form at width {width}"
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/r05_exp1_cascade_table_width.json").write_text(json.dumps(dict(batch="4 x 16 requests, 896 shared + 192 own tokens, Hq 32 / Hkv 8 / D 128",
                                                                                  layers_rotated=L, rows=rows, all_forms_bit_identical=True), indent=1))


if __name__ == "__main__":
    main()
