"""Golden fixtures from the reference's OWN Triton attention kernels, run on an MI355X (test infrastructure).

The logit soft cap and the custom (verify) mask have no CPU-runnable reference implementation: SDPA cannot
express them, so `oracle/ops.py` restates them from the text of the reference's Triton kernels
(kernels/ops/attention/extend_attention.py:409-700 `_fwd_kernel`, decode_attention.py `_fwd_kernel_stage1/2`).
This script pins that restatement: it imports the reference's `extend_attention_fwd` (extend_attention.py:753) and
`decode_attention_fwd` (decode_attention.py:1163) themselves under triton-rocm and records their outputs.

Two steps, because /root/reference does not exist on the GPU box:

    python tests/golden/gen_triton_golden.py --stage      # build container: copies the four kernel files into
                                                          # oracle/_ref/triton_ref/ (git-ignored, never committed,
                                                          # travels with the gpurun snapshot)
    gpurun -- python tests/golden/gen_triton_golden.py    # GPU box: writes gpurun_out/attention_triton.pt
                                                          #          (+ --time: gpurun_out/r03_triton_ref_timing.json)
    cp gpurun_out/attention_triton.pt tests/golden/       # committed fixture

`sglang.srt.environ` / `sglang.srt.utils` are replaced by five-line shims (is_hip() -> True, env switches off);
every line of arithmetic executed is the reference's.  Nothing under tests/ reads oracle/_ref at test time.
"""
from __future__ import annotations

import argparse
import json
import shutil
import sys
import types
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
STAGE = REPO / "oracle" / "_ref" / "triton_ref"
REF = Path("/root/reference/python")
FILES = ["extend_attention.py", "decode_attention.py", "prefill_attention.py", "score_mod.py"]
REL = Path("sglang/kernels/ops/attention")


def stage() -> None:
    if not REF.exists():
        raise SystemExit("/root/reference not present: staging only works in the build container")
    dst = STAGE / REL
    dst.mkdir(parents=True, exist_ok=True)
    for f in FILES:
        shutil.copyfile(REF / REL / f, dst / f)
    print("staged", [str(dst / f) for f in FILES])


def import_reference():
    """The reference modules, imported from the staged copies with shims for the two sglang.srt modules they touch."""
    if not (STAGE / REL / FILES[0]).exists():
        raise SystemExit("run `python tests/golden/gen_triton_golden.py --stage` in the build container first")

    def pkg(name, path=None):
        m = types.ModuleType(name)
        m.__path__ = [str(path)] if path else []
        sys.modules[name] = m
        return m

    pkg("sglang", STAGE / "sglang")
    pkg("sglang.kernels", STAGE / "sglang/kernels")
    pkg("sglang.kernels.ops", STAGE / "sglang/kernels/ops")
    pkg("sglang.kernels.ops.attention", STAGE / REL)
    pkg("sglang.srt")

    class _Env:
        def get(self):
            return False

    class _Envs:
        def __getattr__(self, n):
            return _Env()

    env = pkg("sglang.srt.environ")
    env.envs = _Envs()
    ut = pkg("sglang.srt.utils")
    ut.is_cuda = lambda: False
    ut.is_hip = lambda: True
    ut.is_gfx95_supported = lambda: True
    ut.get_device_core_count = lambda i=0: 256
    import importlib

    ext = importlib.import_module("sglang.kernels.ops.attention.extend_attention")
    dec = importlib.import_module("sglang.kernels.ops.attention.decode_attention")
    return ext, dec


def _layout(torch, g, B_prefix, B_extend, Hkv, D, scale):
    """A scattered page_size=1 pool holding prefix + extend rows of every request."""
    prefix = torch.tensor(B_prefix)
    extend = torch.tensor(B_extend)
    seq = prefix + extend
    slots = int(seq.sum()) + 17
    perm = torch.randperm(slots - 1, generator=g) + 1
    max_ctx = int(seq.max()) + 3
    B = len(B_prefix)
    req_pool = torch.randperm(B, generator=g) + 1
    req_to_token = torch.zeros((B + 1, max_ctx), dtype=torch.int32)
    off = 0
    for i in range(B):
        req_to_token[req_pool[i], : seq[i]] = perm[off: off + seq[i]].to(torch.int32)
        off += int(seq[i])
    k_cache = (torch.randn((slots, Hkv, D), generator=g) * scale).to(torch.bfloat16)
    v_cache = (torch.randn((slots, Hkv, D), generator=g) * 0.5).to(torch.bfloat16)
    return prefix, extend, seq, req_pool, req_to_token, k_cache, v_cache


def _tree_mask(torch, g, n_draft, kv, full_prefix_mask):
    """A verify mask [n_draft, kv + n_draft] of one request: draft token j sees the whole prefix (or a random subset
    of it when `full_prefix_mask`), itself and its ancestors in a random tree over the draft tokens."""
    m = torch.zeros((n_draft, kv + n_draft), dtype=torch.bool)
    m[:, :kv] = True
    if full_prefix_mask:
        m[:, :kv] = torch.rand((n_draft, kv), generator=g) < 0.7
        m[:, 0] = True
    parent = [-1] + [int(torch.randint(0, j, (1,), generator=g)) for j in range(1, n_draft)]
    for j in range(n_draft):
        a = j
        while a >= 0:
            m[j, kv + a] = True
            a = parent[a]
    return m


def run_extend(torch, ext, dev, *, prefix, extend, req_pool, req_to_token, k_cache, v_cache, q, scaling, logit_cap=0.0,
               window=-1, custom_mask=None, mask_indptr=None, skip_prefix_custom_mask=True, is_causal=True):
    """The call of triton_backend.py:1471-1495 (forward_extend) on a page_size=1 pool."""
    B = len(prefix)
    T = int(extend.sum())
    qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
    qo_indptr[1:] = torch.cumsum(extend, 0)
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(prefix, 0)
    kv_indices = torch.cat([req_to_token[req_pool[i], : prefix[i]].to(torch.int64) for i in range(B)] + [torch.zeros(1, dtype=torch.int64)])
    loc = torch.cat([req_to_token[req_pool[i], prefix[i]: prefix[i] + extend[i]].to(torch.int64) for i in range(B)])
    kc, vc = k_cache.to(dev), v_cache.to(dev)
    k_ext, v_ext = kc[loc.to(dev)].contiguous(), vc[loc.to(dev)].contiguous()
    qd = q.to(dev)
    o = torch.empty_like(qd)
    ext.extend_attention_fwd(qd, k_ext, v_ext, o, kc, vc, qo_indptr.to(dev), kv_indptr.to(dev), kv_indices.to(dev),
                             custom_mask.to(dev) if custom_mask is not None else None, is_causal,
                             mask_indptr.to(dev) if mask_indptr is not None else None, int(extend.max()), 1.0, 1.0,
                             scaling, logit_cap=logit_cap, skip_prefix_custom_mask=skip_prefix_custom_mask,
                             sliding_window_size=window)
    torch.cuda.synchronize()
    assert o.shape[0] == T
    return o.cpu()


def run_decode(torch, dec, dev, *, seq, req_pool, req_to_token, k_cache, v_cache, q, scaling, logit_cap=0.0, max_kv_splits=8):
    """The call of triton_backend.py forward_decode (decode_attention_fwd, static kv splits)."""
    B, Hq, D = q.shape
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(seq, 0)
    kv_indices = torch.cat([req_to_token[req_pool[i], : seq[i]].to(torch.int64) for i in range(B)])
    qd = q.to(dev)
    o = torch.empty_like(qd)
    attn_logits = torch.empty((B, Hq, max_kv_splits, D), dtype=torch.float32, device=dev)
    attn_lse = torch.empty((B, Hq, max_kv_splits), dtype=torch.float32, device=dev)
    num_kv_splits = torch.full((B,), max_kv_splits, dtype=torch.int32, device=dev)
    dec.decode_attention_fwd(qd, k_cache.to(dev), v_cache.to(dev), o, kv_indptr.to(dev), kv_indices.to(dev), attn_logits,
                             attn_lse, num_kv_splits, max_kv_splits, scaling, 1.0, 1.0, logit_cap=logit_cap)
    torch.cuda.synchronize()
    return o.cpu()


def generate(out_path: Path) -> None:
    import torch

    ext, dec = import_reference()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(20260923)
    cases = {}
    for name, (Hq, Hkv, D) in {"gqa4_d128": (8, 2, 128), "mha_d64": (4, 4, 64)}.items():
        # K scaled so that q.k * scaling reaches +-40: a cap of 30 bends a good part of the scores
        kscale = 4.0
        pre, extn, seq, req_pool, r2t, kc, vc = _layout(torch, g, [0, 37, 130, 64], [70, 5, 33, 1], Hkv, D, kscale)
        T = int(extn.sum())
        q = (torch.randn((T, Hq, D), generator=g) * 1.5).to(torch.bfloat16)
        scaling = D ** -0.5
        base = dict(prefix=pre, extend=extn, req_pool=req_pool, req_to_token=r2t, k_cache=kc, v_cache=vc, q=q, scaling=scaling)
        c = dict(q=q, k_cache=kc, v_cache=vc, req_to_token=r2t, req_pool_indices=req_pool, seq_lens=seq,
                 extend_prefix_lens=pre, extend_seq_lens=extn, scaling=scaling, logit_cap=30.0, sliding_window=16)
        c["out_extend_causal"] = run_extend(torch, ext, dev, **base)
        c["out_extend_cap"] = run_extend(torch, ext, dev, **base, logit_cap=30.0)
        c["out_extend_window"] = run_extend(torch, ext, dev, **base, window=16)
        c["out_extend_cap_window"] = run_extend(torch, ext, dev, **base, logit_cap=30.0, window=16)
        qd = (torch.randn((len(seq), Hq, D), generator=g) * 1.5).to(torch.bfloat16)
        dbase = dict(seq=seq, req_pool=req_pool, req_to_token=r2t, k_cache=kc, v_cache=vc, q=qd, scaling=scaling)
        c["q_decode"] = qd
        c["out_decode"] = run_decode(torch, dec, dev, **dbase)
        c["out_decode_cap"] = run_decode(torch, dec, dev, **dbase, logit_cap=30.0)
        # TARGET_VERIFY (triton_backend.py:860-919): every request extends by n_draft tokens over its whole context,
        # mask [n_draft, seq + n_draft] per request, flattened and concatenated; the prefix part of the mask is skipped
        # (skip_prefix_custom_mask, extend_attention.py:774) in the backend's call
        n_draft = 8
        vpre = torch.tensor([45, 130, 7, 64])
        vext = torch.full((4,), n_draft)
        _, _, vseq, vpool, vr2t, vkc, vvc = _layout(torch, g, vpre.tolist(), vext.tolist(), Hkv, D, kscale)
        vq = (torch.randn((4 * n_draft, Hq, D), generator=g) * 1.5).to(torch.bfloat16)
        for tag, full in (("verify", False), ("verify_prefix_masked", True)):
            masks = [_tree_mask(torch, g, n_draft, int(vpre[i]), full) for i in range(4)]
            flat = torch.cat([m.flatten() for m in masks])
            mip = torch.zeros(5, dtype=torch.int64)
            mip[1:] = torch.cumsum(torch.tensor([m.numel() for m in masks]), 0)
            vb = dict(prefix=vpre, extend=vext, req_pool=vpool, req_to_token=vr2t, k_cache=vkc, v_cache=vvc, q=vq, scaling=scaling,
                      custom_mask=flat, mask_indptr=mip, skip_prefix_custom_mask=not full)
            c[f"{tag}_mask"] = flat
            c[f"{tag}_mask_indptr"] = mip
            c[f"out_{tag}"] = run_extend(torch, ext, dev, **vb)
            c[f"out_{tag}_cap"] = run_extend(torch, ext, dev, **vb, logit_cap=30.0)
        c.update(verify=dict(q=vq, k_cache=vkc, v_cache=vvc, req_to_token=vr2t, req_pool_indices=vpool, seq_lens=vseq,
                             extend_prefix_lens=vpre, extend_seq_lens=vext, n_draft=n_draft))
        cases[name] = c
    import triton

    cases["_meta"] = dict(triton=triton.__version__, torch=str(torch.__version__), device=torch.cuda.get_device_name(0),
                          source="reference extend_attention_fwd / decode_attention_fwd, staged copies")
    out_path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(cases, out_path)
    print("wrote", out_path, {k: len(v) for k, v in cases.items()})


def time_reference(out_path: Path) -> None:
    """The reference's ROCm attention kernels on the benchmark's shapes (Llama-3-8B, B=64): cold prefill 4 x 1024,
    warm prefill 60 x 128 over 896, decode 64 x 1088 -- context for the product's own timings, not a target."""
    import torch

    ext, dec = import_reference()
    dev = torch.device("cuda", 0)
    Hq, Hkv, D = 32, 8, 128
    res = {}

    def timed(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    for tag, (B, pre, extn) in {"cold": (4, 0, 1024), "warm": (60, 896, 128), "long": (2, 0, 4096)}.items():
        seq = pre + extn
        slots = B * seq + 1
        kc = (torch.randn((slots, Hkv, D), device=dev) * 0.5).to(torch.bfloat16)
        vc = (torch.randn((slots, Hkv, D), device=dev) * 0.5).to(torch.bfloat16)
        q = (torch.randn((B * extn, Hq, D), device=dev) * 0.5).to(torch.bfloat16)
        o = torch.empty_like(q)
        base = torch.arange(B, device=dev)[:, None] * seq + 1
        kv_indices = (base + torch.arange(pre, device=dev)[None, :]).flatten().to(torch.int64)
        loc = (base + torch.arange(pre, seq, device=dev)[None, :]).flatten()
        k_ext, v_ext = kc[loc].contiguous(), vc[loc].contiguous()
        qo = torch.arange(0, (B + 1) * extn, extn, dtype=torch.int32, device=dev)
        kvp = torch.arange(0, (B + 1) * pre, max(pre, 1), dtype=torch.int32, device=dev) if pre else torch.zeros(B + 1, dtype=torch.int32, device=dev)
        kv_indices = torch.cat([kv_indices, torch.zeros(1, dtype=torch.int64, device=dev)])
        us = timed(lambda: ext.extend_attention_fwd(q, k_ext, v_ext, o, kc, vc, qo, kvp, kv_indices, None, True, None, extn,
                                                    1.0, 1.0, D ** -0.5))
        flops = 4.0 * Hq * D * B * (extn * pre + extn * (extn + 1) / 2)
        res[f"extend_{tag}"] = dict(us=us, tflops=flops / us / 1e6, shape=dict(requests=B, extend=extn, prefix=pre))
    B, seq = 64, 1088
    slots = B * seq + 1
    kc = (torch.randn((slots, Hkv, D), device=dev) * 0.5).to(torch.bfloat16)
    vc = (torch.randn((slots, Hkv, D), device=dev) * 0.5).to(torch.bfloat16)
    q = (torch.randn((B, Hq, D), device=dev) * 0.5).to(torch.bfloat16)
    o = torch.empty_like(q)
    kv_indptr = torch.arange(0, (B + 1) * seq, seq, dtype=torch.int32, device=dev)
    kv_indices = (torch.arange(B * seq, device=dev) + 1).to(torch.int64)
    for splits in (8, 16):
        al = torch.empty((B, Hq, splits, D), dtype=torch.float32, device=dev)
        ls = torch.empty((B, Hq, splits), dtype=torch.float32, device=dev)
        nk = torch.full((B,), splits, dtype=torch.int32, device=dev)
        us = timed(lambda: dec.decode_attention_fwd(q, kc, vc, o, kv_indptr, kv_indices, al, ls, nk, splits, D ** -0.5, 1.0, 1.0))
        byts = B * seq * 2 * Hkv * D * 2
        res[f"decode_splits{splits}"] = dict(us=us, gbps=byts / us / 1e3, bytes_no_dedup=byts)
    import triton

    res["_meta"] = dict(triton=triton.__version__, device=torch.cuda.get_device_name(0),
                        note="reference Triton kernels (extend_attention_fwd / decode_attention_fwd), HIP block sizes of the "
                             "reference's own _get_block_sizes_for_extend_attention; radix prefix sharing does not change "
                             "their traffic: the decode kernel reads every request's whole context")
    out_path.write_text(json.dumps(res, indent=1))
    print(json.dumps(res))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", action="store_true")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "attention_triton.pt"))
    a = ap.parse_args()
    if a.stage:
        stage()
    elif a.time:
        time_reference(REPO / "gpurun_out" / "r03_triton_ref_timing.json")
    else:
        generate(Path(a.out))
