"""SURVEY.md section 8 (f3) / (f4) on the GPU: the attention and store kernels over the pool formats and masks beyond
the bf16 NHD default -- OCP e4m3 KV rows with scales, HND paged layout, sliding window, logit soft cap, and the
custom (tree / verify) mask of the extend kernel -- each against the oracle's restatement (oracle/ops.py; the window
semantics pinned to the real reference by tests/test_oracle_golden.py)."""
import dataclasses
import random

import pytest
import torch

from oracle import ops as oo
from oracle.model import OracleLM, weights_from_product_model

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _k():
    from sglang_amd import kernels

    return kernels


def _pool(slots, Hkv, D, page, hnd, fp8, seed):
    """bf16 source rows + the pool tensors in the requested format (filled through the product's store kernel)."""
    g = torch.Generator().manual_seed(seed)
    k = (torch.randn((slots, Hkv, D), generator=g) * 0.6).to(BF)
    v = (torch.randn((slots, Hkv, D), generator=g) * 0.6).to(BF)
    return k, v


def _store(device, k_rows, v_rows, page, hnd, fp8, k_scale, v_scale):
    K = _k()
    slots, Hkv, D = k_rows.shape
    dt = torch.uint8 if fp8 else BF
    shape = (slots // page, Hkv, page, D) if hnd else (slots, Hkv, D)
    kc = torch.zeros(shape, dtype=dt, device=device)
    vc = torch.zeros(shape, dtype=dt, device=device)
    loc = torch.arange(slots, dtype=torch.int64, device=device)
    K.store_kv_cache(k_rows.to(device), v_rows.to(device), kc, vc, loc, num_kv_heads=Hkv, head_dim=D, kv_fp8=fp8,
                     k_scale=k_scale, v_scale=v_scale, page_size=page, hnd=hnd)
    return kc, vc


@pytest.mark.parametrize("page,hnd,fp8", [(1, False, True), (4, True, False), (16, True, True)])
def test_store_kv_formats_bit_exact(device, page, hnd, fp8):
    Hkv, D, slots = 2, 128, 64
    k, v = _pool(slots, Hkv, D, page, hnd, fp8, 3)
    ks, vs = (0.5, 2.0) if fp8 else (1.0, 1.0)
    kc, vc = _store(device, k, v, page, hnd, fp8, ks, vs)
    for rows, cache, sc in ((k, kc, ks), (v, vc, vs)):
        want = oo.quantize_kv_fp8(rows, sc).view(torch.uint8) if fp8 else rows
        got = cache.cpu()
        if hnd:                                           # [pages, H, page, D] -> [slots, H, D]
            got = got.permute(0, 2, 1, 3).reshape(slots, Hkv, D)
        assert torch.equal(got, want)


def _batch(lens, slots, ctx, seed):
    rnd = random.Random(seed)
    B = len(lens)
    r2t = torch.zeros((B + 1, ctx), dtype=torch.int32)
    perm = list(range(1, slots)); rnd.shuffle(perm)
    off = 0
    for b, n in enumerate(lens):
        r2t[b + 1, :n] = torch.tensor(perm[off: off + n], dtype=torch.int32); off += n
    return r2t, torch.arange(1, B + 1)


CASES = [dict(fp8=True), dict(hnd=True, page=4), dict(hnd=True, page=16, fp8=True), dict(window=40), dict(cap=30.0),
         dict(fp8=True, window=17, cap=50.0), dict(window=0)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
@pytest.mark.parametrize("Hq,Hkv,D", [(8, 2, 128), (4, 4, 64)])
def test_decode_attention_formats(device, case, Hq, Hkv, D):
    K = _k()
    fp8, hnd, page = case.get("fp8", False), case.get("hnd", False), case.get("page", 1)
    window, cap = case.get("window", -1), case.get("cap", 0.0)
    ks, vs = (0.5, 1.5) if fp8 else (1.0, 1.0)
    lens = [1, 37, 130, 300, 64]
    slots = 1024
    k, v = _pool(slots, Hkv, D, page, hnd, fp8, 5)
    kc, vc = _store(device, k, v, page, hnd, fp8, ks, vs)
    r2t, pool = _batch(lens, slots, 320, 7)
    q = (torch.randn((len(lens), Hq, D), generator=torch.Generator().manual_seed(1)) * 0.5).to(BF)
    seq = torch.tensor(lens, dtype=torch.int32)
    out = torch.empty((len(lens), Hq, D), dtype=BF, device=device)
    for splits in (1, 2):
        ws = K.decode_workspace(len(lens), Hq, D, splits, device) if splits > 1 else (None, None)
        K.decode_attention(q.to(device), kc, vc, out, r2t.to(device), pool.to(device), seq.to(device), D ** -0.5, splits, ws[0], ws[1],
                           kv_fp8=fp8, k_scale=ks, v_scale=vs, page_size=page, hnd=hnd, sliding_window=window, logit_cap=cap)
        kref = oo.quantize_kv_fp8(k, ks) if fp8 else k
        vref = oo.quantize_kv_fp8(v, vs) if fp8 else v
        ref = oo.decode_attention(q, kref, vref, r2t, pool, seq.long(), D ** -0.5, compute_dtype=torch.float32, k_scale=ks, v_scale=vs,
                                  sliding_window=window, logit_cap=cap)
        err = (out.cpu().float() - ref.float()).abs()
        assert float(err.max()) <= 2.0 ** -7 * float(ref.float().abs().max()) + 2e-3, (splits, float(err.max()))


@pytest.mark.parametrize("case", [dict(fp8=True), dict(hnd=True, page=4), dict(hnd=True, page=16, fp8=True)],
                         ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
@pytest.mark.parametrize("Hq,Hkv,D", [(8, 2, 128), (4, 4, 64)])
def test_cascade_decode_attention_formats(device, case, Hq, Hkv, D):
    """The shared-prefix kernel over fp8 / paged head-major pools: two groups sharing 256 / 128 cached tokens plus a
    loner, against the oracle's per-request attention on the same (quantised) rows."""
    K = _k()
    fp8, hnd, page = case.get("fp8", False), case.get("hnd", False), case.get("page", 1)
    ks, vs = (0.5, 1.5) if fp8 else (1.0, 1.0)
    slots, ctx = 4096, 640
    k, v = _pool(slots, Hkv, D, page, hnd, fp8, 11)
    kc, vc = _store(device, k, v, page, hnd, fp8, ks, vs)
    lens = [300, 333, 290, 513, 200, 170, 77]
    groups = [(0, 1, 2, 3), (4, 5)]                      # members of a group share their first `shared` slots
    shared = {0: 256, 1: 128}
    r2t, pool = _batch(lens, slots, ctx, 13)
    for gi, members in enumerate(groups):
        for m in members[1:]:
            r2t[m + 1, :shared[gi]] = r2t[members[0] + 1, :shared[gi]]
    B = len(lens)
    q = (torch.randn((B, Hq, D), generator=torch.Generator().manual_seed(2)) * 0.5).to(BF)
    seq = torch.tensor(lens, dtype=torch.int32)
    out = torch.empty((B, Hq, D), dtype=BF, device=device)
    ws = K.CascadeWorkspace(B, Hq, D, ctx, device)
    K.cascade_plan(ws, r2t.to(device), pool.to(device), seq.to(device), Hq, Hkv)
    summ = K.cascade_plan_summary(ws, B)
    assert summ["n_groups"] == 2, summ
    K.cascade_decode_attention(ws, q.to(device), kc, vc, out, r2t.to(device), pool.to(device), seq.to(device), D ** -0.5,
                               kv_fp8=fp8, k_scale=ks, v_scale=vs, page_size=page, hnd=hnd)
    kref = oo.quantize_kv_fp8(k, ks) if fp8 else k
    vref = oo.quantize_kv_fp8(v, vs) if fp8 else v
    ref = oo.decode_attention(q, kref, vref, r2t, pool, seq.long(), D ** -0.5, compute_dtype=torch.float32, k_scale=ks, v_scale=vs)
    err = (out.cpu().float() - ref.float()).abs()
    assert float(err.max()) <= 2.0 ** -7 * float(ref.float().abs().max()) + 2e-3, float(err.max())


@pytest.mark.parametrize("case", CASES + [dict(mask=True), dict(mask=True, fp8=True)], ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_extend_attention_formats(device, case, extend_shape):
    K = _k()
    # the small grid of this batch would pick a 4-wave shape: force the 8-wave one so that the token-major layouts (bf16
    # and e4m3 rows) go through the 32x32 kernel (HND / window / cap / mask cases stay on the general kernel by construction)
    extend_shape("82")
    Hq, Hkv, D = 8, 2, 128
    fp8, hnd, page = case.get("fp8", False), case.get("hnd", False), case.get("page", 1)
    window, cap, use_mask = case.get("window", -1), case.get("cap", 0.0), case.get("mask", False)
    ks, vs = (0.5, 1.5) if fp8 else (1.0, 1.0)
    prefix = [0, 70, 200, 5]
    extend = [150, 33, 1, 64]
    lens = [p + e for p, e in zip(prefix, extend)]
    slots = 1024
    k, v = _pool(slots, Hkv, D, page, hnd, fp8, 9)
    kc, vc = _store(device, k, v, page, hnd, fp8, ks, vs)
    r2t, pool = _batch(lens, slots, 320, 11)
    T = sum(extend)
    g = torch.Generator().manual_seed(4)
    q = (torch.randn((T, Hq, D), generator=g) * 0.5).to(BF)
    seq, pre, ext = (torch.tensor(x, dtype=torch.int32) for x in (lens, prefix, extend))
    qo = torch.zeros(len(lens) + 1, dtype=torch.int32); qo[1:] = torch.cumsum(ext, 0)
    mask = indptr = None
    if use_mask:
        # a tree-verify style mask: every query sees the whole prefix, a random subset of the extend part and itself
        blocks, sizes = [], []
        for p_, e_ in zip(prefix, extend):
            m = torch.zeros((e_, p_ + e_), dtype=torch.bool)
            m[:, :p_] = True
            m[:, p_:] = torch.rand((e_, e_), generator=g) < 0.4
            m[torch.arange(e_), p_ + torch.arange(e_)] = True
            m[-1, :] = False if e_ > 2 else m[-1, :]          # one query row that may attend to nothing
            blocks.append(m.flatten()); sizes.append(m.numel())
        mask = torch.cat(blocks)
        indptr = torch.zeros(len(lens) + 1, dtype=torch.int64); indptr[1:] = torch.cumsum(torch.tensor(sizes), 0)
    out = torch.empty((T, Hq, D), dtype=BF, device=device)
    K.extend_attention(q.to(device), out, kc, vc, r2t.to(device), pool.to(device), seq.to(device), pre.to(device), qo.to(device),
                       max(extend), D ** -0.5, True, kv_fp8=fp8, k_scale=ks, v_scale=vs, page_size=page, hnd=hnd,
                       sliding_window=window, logit_cap=cap, custom_mask=mask.to(device) if use_mask else None,
                       mask_indptr=indptr.to(device) if use_mask else None)
    kref = oo.quantize_kv_fp8(k, ks) if fp8 else k
    vref = oo.quantize_kv_fp8(v, vs) if fp8 else v
    ref = oo.extend_attention(q, kref, vref, r2t, pool, seq.long(), pre.long(), ext.long(), D ** -0.5, True, torch.float32,
                              k_scale=ks, v_scale=vs, sliding_window=window, logit_cap=cap, custom_mask=mask,
                              mask_indptr=indptr.tolist() if use_mask else None)
    err = (out.cpu().float() - ref.float()).abs()
    assert float(err.max()) <= 2.0 ** -7 * float(ref.float().abs().max()) + 4e-3, float(err.max())


@pytest.mark.parametrize("kv,hnd,page,window,cap", [("fp8_e4m3", False, 1, None, 0.0), ("auto", True, 4, None, 0.0),
                                                    ("fp8_e4m3", True, 16, None, 0.0), ("auto", False, 1, 24, 0.0),
                                                    ("auto", False, 1, None, 20.0)])
def test_engine_with_pool_formats_and_layer_switches(device, kv, hnd, page, window, cap):
    """The whole path (radix-cached prefill, hipGraph decode) over an fp8 / HND pool, with sliding-window or
    soft-capped layers: logits against the oracle model run with the same pool format / switches."""
    from sglang_amd.harness.engine import Engine, ModelRunner, Req
    from sglang_amd.harness.models import CONFIGS

    cfg = dataclasses.replace(CONFIGS["tiny-llama"], sliding_window=window, logit_cap=cap, name="tiny-llama-fmt")
    runner = ModelRunner(cfg, max_total_tokens=2048, max_running_requests=8, max_context_len=256, page_size=page, device=device,
                         init_device="cpu", use_graph=True, kv_cache_dtype=kv, use_hnd=hnd)
    eng = Engine(runner)
    rnd = random.Random(2)
    shared = [rnd.randrange(cfg.vocab_size) for _ in range(48)]
    prompts = [shared + [rnd.randrange(cfg.vocab_size) for _ in range(9)] for _ in range(4)]
    new_tokens = 6
    eng.logits_by_req = {}
    reqs = [Req(i, p, new_tokens) for i, p in enumerate(prompts)]
    eng.prefill(reqs[:1]); eng.prefill(reqs[1:])
    assert all(q.cached_tokens == 48 // page * page for q in reqs[1:])
    for _ in range(new_tokens - 1):
        eng.decode_step()
    eng.finish(list(eng.running))
    oracle = OracleLM(cfg, weights_from_product_model(runner.model), compute_dtype=torch.float32, kv_cache_dtype=kv)
    _, ref = oracle.generate(prompts, new_tokens, return_logits=True, forced=[q.output_ids for q in reqs])
    for b, q in enumerate(reqs):
        for k_, row in enumerate(eng.logits_by_req[q.rid]):
            torch.testing.assert_close(row, ref[k_][b], atol=3e-2, rtol=3e-2, msg=f"request {q.rid} token {k_}")
