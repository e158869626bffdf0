"""GPU parity tests: every HIP kernel (through the C ABI) against the CPU oracle
and the committed golden fixtures.  Run with `pytest -m gpu` on an MI355X."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import host as oh
from oracle import ops as oo

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _k():
    from sglang_amd import kernels

    return kernels


def _load(golden_dir, name):
    return torch.load(golden_dir / name, weights_only=False)


def _ulp_close(a: torch.Tensor, b: torch.Tensor, max_frac: float = 0.0, atol_ulps: int = 1):
    """bf16 tensors equal up to `atol_ulps` bf16 ulps; at most max_frac of elements may differ at all."""
    a16 = a.cpu().view(torch.int16).to(torch.int32)
    b16 = b.cpu().view(torch.int16).to(torch.int32)
    # map sign-magnitude to a monotone integer line
    def mono(x):
        return torch.where(x < 0, -(x & 0x7FFF), x)
    d = (mono(a16) - mono(b16)).abs()
    assert int(d.max()) <= atol_ulps, f"max bf16 ulp diff {int(d.max())}"
    frac = float((d > 0).float().mean())
    assert frac <= max_frac, f"{frac:.4%} elements differ (allowed {max_frac:.4%})"


# ------------------------------------------------------------------ MFMA lane map
def test_mfma_probe_layout(device):
    g = torch.Generator().manual_seed(3)
    a = torch.randn((16, 32), generator=g).to(BF)
    b = torch.randn((32, 16), generator=g).to(BF)   # asymmetric on purpose
    c = _k().probe_mfma_16x16x32(a.to(device), b.to(device)).cpu()
    ref = a.float() @ b.float()
    torch.testing.assert_close(c, ref, atol=1e-3, rtol=1e-3)


# ------------------------------------------------------------------ elementwise
def test_rmsnorm_golden(device, golden_dir):
    g = _load(golden_dir, "elementwise_native.pt")
    for name, c in g.items():
        if not name.startswith("rmsnorm"):
            continue
        out = _k().rmsnorm(c["x"].to(device), c["weight"].to(device), c["eps"])
        _ulp_close(out, c["out"], max_frac=0.002)
        x = c["x"].to(device).clone()
        r = c["residual"].to(device).clone()
        _k().fused_add_rmsnorm(x, r, c["weight"].to(device), c["eps"])
        assert torch.equal(r.cpu(), c["residual_out"]), "residual update must be bit exact"
        _ulp_close(x, c["out_fused"], max_frac=0.002)


@pytest.mark.parametrize("shape", [(1, 4096), (128, 4096), (7, 2048), (33, 8192), (5, 896)])
def test_rmsnorm_shapes(device, shape):
    # shapes of test/registered/kernels/ops/layernorm/test_fused_op_gpu_parity.py:37-130
    torch.manual_seed(0)
    x = torch.randn(shape).to(BF)
    r = torch.randn(shape).to(BF)
    w = (1 + 0.1 * torch.randn(shape[-1])).to(BF)
    out = _k().rmsnorm(x.to(device), w.to(device), 1e-6)
    _ulp_close(out, oo.rmsnorm(x, w, 1e-6), max_frac=0.003)
    xd, rd = x.to(device).clone(), r.to(device).clone()
    _k().fused_add_rmsnorm(xd, rd, w.to(device), 1e-6)
    y, r2 = oo.fused_add_rmsnorm(x, r, w, 1e-6)
    assert torch.equal(rd.cpu(), r2)
    _ulp_close(xd, y, max_frac=0.003)


def test_rmsnorm_strided_rows(device):
    torch.manual_seed(1)
    big = torch.randn((9, 3 * 1024)).to(BF)
    w = torch.ones(1024).to(BF)
    xs = big.to(device)[:, 1024:2048]          # row stride 3072
    out = _k().rmsnorm(xs, w.to(device), 1e-5)
    _ulp_close(out, oo.rmsnorm(big[:, 1024:2048], w, 1e-5), max_frac=0.003)


@pytest.mark.parametrize("shape", [(1, 2 * 14336), (64, 2 * 14336), (7, 2 * 4864), (3, 2 * 3584)])
def test_silu_and_mul(device, shape):
    torch.manual_seed(0)
    x = torch.randn(shape).to(BF)
    out = _k().silu_and_mul(x.to(device))
    _ulp_close(out, oo.silu_and_mul(x), max_frac=0.002)


def test_rope_golden(device, golden_dir):
    g = _load(golden_dir, "elementwise_native.pt")
    for name, c in g.items():
        if not name.startswith("rope"):
            continue
        for cache in (c["cache"], c["cache_f32"]):
            q, k = c["q"].to(device).clone(), c["k"].to(device).clone()
            _k().rotary_embedding(c["positions"].to(device), q, k, c["head_size"], cache.to(device), c["is_neox"])
            assert torch.equal(q.cpu(), c["q_out"]), name
            assert torch.equal(k.cpu(), c["k_out"]), name


def test_rope_fused_store_and_store_kv(device):
    torch.manual_seed(2)
    T, Hq, Hk, D, slots = 37, 8, 2, 128, 101
    cache = oo.cos_sin_cache(oo.rope_inv_freq(D, 10000.0), 300).to(BF)
    pos = torch.randint(0, 300, (T,))
    q = torch.randn((T, Hq * D)).to(BF)
    k = torch.randn((T, Hk * D)).to(BF)
    v = torch.randn((T, Hk * D)).to(BF)
    loc = (torch.randperm(slots - 1)[:T] + 1)
    qo, ko = oo.rotary_embedding(pos, q, k, D, cache, True)
    kc = torch.zeros((slots, Hk, D), dtype=BF)
    vc = torch.zeros((slots, Hk, D), dtype=BF)
    oo.store_kv(ko, v, kc, vc, loc)
    qd, kd = q.to(device).clone(), k.to(device).clone()
    kcd = torch.zeros((slots, Hk, D), dtype=BF, device=device)
    vcd = torch.zeros((slots, Hk, D), dtype=BF, device=device)
    _k().rotary_embedding(pos.to(device), qd, kd, D, cache.to(device), True, value=v.to(device), k_cache=kcd,
                          v_cache=vcd, cache_loc=loc.to(device))
    assert torch.equal(qd.cpu(), qo) and torch.equal(kd.cpu(), ko)
    assert torch.equal(kcd.cpu(), kc) and torch.equal(vcd.cpu(), vc)
    kcd.zero_(); vcd.zero_()
    _k().store_kv_cache(kd, v.to(device), kcd, vcd, loc.to(device))
    assert torch.equal(kcd.cpu(), kc) and torch.equal(vcd.cpu(), vc)


# ------------------------------------------------------------------ integer metadata (bit exact)
@pytest.mark.parametrize("batch", [1, 37, 1786])
def test_create_kv_indices_reference_test(device, batch):
    # test/registered/attention/test_create_kvindices.py:24-71
    rng = np.random.default_rng(batch)
    max_batch, max_ctx = 4096, 4096
    r2t = torch.arange(max_batch * max_ctx, dtype=torch.int32).reshape(max_batch, max_ctx)
    pool = torch.from_numpy(rng.choice(max_batch, size=batch, replace=False)).to(torch.int32)
    lens = torch.from_numpy(rng.choice(max_ctx, size=batch, replace=False)).to(torch.int32)
    indptr, ref = oh.create_kv_indices(r2t.numpy(), pool.numpy(), lens.numpy())
    for pool_dtype in (torch.int32, torch.int64):
        for out_dtype in (torch.int32, torch.int64):
            out = torch.empty(int(indptr[-1]), dtype=out_dtype, device=device)
            _k().create_kv_indices(r2t.to(device), pool.to(pool_dtype).to(device), lens.to(device),
                                   torch.from_numpy(indptr).to(device), None, out)
            assert torch.equal(out.cpu().to(torch.int64), torch.from_numpy(ref).to(torch.int64))


def test_host_int_golden(device, golden_dir):
    g = json.loads((golden_dir / "host_int.json").read_text())
    for c in g["alloc_extend"]:
        out = torch.full((len(c["out"]),), -7, dtype=torch.int64, device=device)
        t = lambda x: torch.tensor(x, dtype=torch.int64, device=device)
        _k().alloc_extend(t(c["prefix"]), t(c["seq"]), t(c["last_loc"]), t(c["free_pages"]), out, c["page_size"])
        assert out.cpu().tolist() == c["out"], c
    for c in g["compute_position"]:
        for dt in (torch.int32, torch.int64):
            p, s = _k().compute_position(torch.tensor(c["prefix"], dtype=dt, device=device),
                                         torch.tensor(c["extend"], dtype=dt, device=device), sum(c["extend"]))
            assert p.cpu().tolist() == c["positions"] and s.cpu().tolist() == c["start"]
    c = g["get_last_loc"]
    r2t = torch.arange(5 * 11, dtype=torch.int32).reshape(5, 11).to(device)
    ll = _k().get_last_loc(r2t, torch.tensor(c["req_pool"], device=device), torch.tensor(c["prefix"], device=device))
    assert ll.cpu().tolist() == c["out"]
    c = g["clamp_position"]
    assert _k().clamp_position(torch.tensor(c["seq"], device=device)).cpu().tolist() == c["out"]


def test_alloc_decode_and_write_req_to_token(device):
    rng = np.random.default_rng(5)
    for page_size in (1, 4, 16):
        bs = 9
        seq = rng.integers(1, 60, size=bs)
        last_loc = [(1000 + i) * page_size + (int(s) - 2) % page_size if s > 1 else -1 for i, s in enumerate(seq)]
        free_pages = rng.choice(np.arange(1, 500), size=bs + 2, replace=False)
        ref, _ = oh.alloc_decode(seq, last_loc, free_pages, page_size)
        out = torch.empty(bs, dtype=torch.int64, device=device)
        t = lambda x: torch.tensor(np.asarray(x), dtype=torch.int64, device=device)
        _k().alloc_decode(t(seq), t(last_loc), t(free_pages), out, page_size)
        assert out.cpu().tolist() == ref.tolist()
    # write_req_to_token (allocation.py:85-103 CPU loop is the spec)
    bs, ctx = 5, 64
    prefix = [0, 3, 10, 0, 7]
    ext = [4, 1, 9, 2, 5]
    seq = [p + e for p, e in zip(prefix, ext)]
    pool = [3, 1, 5, 2, 4]
    prefix_tensors = [torch.arange(100 * (i + 1), 100 * (i + 1) + p, dtype=torch.int64, device=device)
                      for i, p in enumerate(prefix)]
    out_loc = torch.arange(500, 500 + sum(ext), dtype=torch.int64, device=device)
    r2t = torch.zeros((bs + 1, ctx), dtype=torch.int32, device=device)
    ptrs = torch.tensor([p.data_ptr() for p in prefix_tensors], dtype=torch.int64, device=device)
    t = lambda x: torch.tensor(x, dtype=torch.int64, device=device)
    _k().write_req_to_token(r2t, t(pool), ptrs, t(prefix), t(seq), t(ext), out_loc)
    ref = torch.zeros((bs + 1, ctx), dtype=torch.int32)
    pt = 0
    for i in range(bs):
        ref[pool[i], :prefix[i]] = prefix_tensors[i].cpu().to(torch.int32)
        ref[pool[i], prefix[i]:seq[i]] = out_loc[pt:pt + ext[i]].cpu().to(torch.int32)
        pt += ext[i]
    assert torch.equal(r2t.cpu(), ref)


# ------------------------------------------------------------------ attention
def _to_dev(c, device, keys):
    return {k: c[k].to(device) for k in keys}


def _run_decode(c, device, q, num_splits=1, flags=0):
    k = _k()
    B, Hq, D = q.shape
    out = torch.empty_like(q, device=device)
    ws = k.decode_workspace(B, Hq, D, num_splits, device) if num_splits > 1 else (None, None)
    k.decode_attention(q.to(device), c["k_cache"].to(device), c["v_cache"].to(device), out,
                       c["req_to_token"].to(device), c["req_pool_indices"].to(torch.int64).to(device),
                       c["seq_lens"].to(torch.int32).to(device), c["scaling"], num_splits, ws[0], ws[1], flags=flags)
    return out.cpu()


def _run_extend(c, device, causal=True):
    k = _k()
    q = c["q"]
    out = torch.empty_like(q, device=device)
    ext = c["extend_seq_lens"].to(torch.int32)
    qo = torch.zeros(len(ext) + 1, dtype=torch.int32)
    qo[1:] = torch.cumsum(ext, 0)
    k.extend_attention(q.to(device), out, c["k_cache"].to(device), c["v_cache"].to(device),
                       c["req_to_token"].to(device), c["req_pool_indices"].to(torch.int64).to(device),
                       c["seq_lens"].to(torch.int32).to(device), c["extend_prefix_lens"].to(torch.int32).to(device),
                       qo.to(device), int(ext.max()), c["scaling"], causal)
    return out.cpu()


def test_decode_advance_equals_the_eager_ops(device):
    """One launch = the decode step's five eager tensor ops (allocation.py:512-560 alloc_for_decode +
    write_req_to_token_pool at column seq_len, then seq_lens += 1), for a few steps in a row."""
    bs, ctx = 37, 96
    g = torch.Generator().manual_seed(3)
    pool = torch.randperm(64, generator=g)[:bs].to(torch.int64) + 1
    seq0 = torch.randint(1, 80, (bs,), generator=g, dtype=torch.int32)
    r2t_a = torch.randint(0, 1000, (70, ctx), generator=g, dtype=torch.int32)
    r2t_b = r2t_a.clone().to(device)
    seq_a, seq_b = seq0.clone(), seq0.clone().to(device)
    out_b = torch.zeros(bs, dtype=torch.int64, device=device)
    for step in range(4):
        slots = (torch.randperm(5000, generator=g)[:bs] + 10 + 10000 * step).to(torch.int64)
        # the eager form (harness/engine.py, CPU path)
        r2t_a[pool, seq_a.long()] = slots.to(torch.int32)
        seq_a += 1
        _k().decode_advance(r2t_b, pool.to(device), seq_b, slots.to(device), out_b)
        assert torch.equal(out_b.cpu(), slots)
        assert torch.equal(seq_b.cpu(), seq_a)
        assert torch.equal(r2t_b.cpu(), r2t_a)


def test_attention_golden(device, golden_dir):
    """Fixtures recorded from the real TorchNativeAttnBackend (bf16 SDPA)."""
    cases = _load(golden_dir, "attention_torch_native.pt")
    for name, c in cases.items():
        for flags in (0, 1):
            for splits in (1, 3):
                o = _run_decode(c, device, c["q_decode"], splits, flags)
                torch.testing.assert_close(o.float(), c["out_decode"].float(), atol=1e-2, rtol=1e-2, msg=f"{name} decode")
        o = _run_extend(c, device)
        torch.testing.assert_close(o.float(), c["out_extend"].float(), atol=1e-2, rtol=1e-2, msg=f"{name} extend")


def _random_case(B, Hq, Hkv, D, prefix, extend, seed, slots=None, spike=False):
    g = torch.Generator().manual_seed(seed)
    prefix = torch.tensor(prefix)
    extend = torch.tensor(extend)
    seq = prefix + extend
    total = int(seq.sum())
    slots = slots or total + 17
    max_ctx = int(seq.max()) + 3
    perm = torch.randperm(slots - 1, generator=g)[:total] + 1
    r2t = torch.zeros((B + 2, max_ctx), dtype=torch.int32)
    pool = torch.randperm(B + 1, generator=g)[:B] + 1
    off = 0
    for i in range(B):
        r2t[pool[i], : seq[i]] = perm[off: off + seq[i]].to(torch.int32)
        off += int(seq[i])
    # inputs like test/registered/attention/test_triton_attention_kernels.py (normal(0.1, 0.2))
    kc = (torch.randn((slots, Hkv, D), generator=g) * 0.2 + 0.1).to(BF)
    vc = (torch.randn((slots, Hkv, D), generator=g) * 0.2 + 0.1).to(BF)
    q = (torch.randn((int(extend.sum()), Hq, D), generator=g) * 0.2 + 0.1).to(BF)
    qd = (torch.randn((B, Hq, D), generator=g) * 0.2 + 0.1).to(BF)
    if spike:  # force the online-softmax rescale path: one key dominates late in the sequence
        kc[perm[total - 3]] = (qd[B - 1, 0] * 40).to(BF)
    return dict(q=q, q_decode=qd, k_cache=kc, v_cache=vc, req_to_token=r2t, req_pool_indices=pool, seq_lens=seq,
                extend_prefix_lens=prefix, extend_seq_lens=extend, scaling=D ** -0.5)


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (14, 2, 64), (16, 4, 128), (4, 4, 64), (12, 4, 256)])
def test_decode_attention_random(device, Hq, Hkv, D):
    lens = [1, 2, 63, 64, 65, 127, 128, 129, 300, 1153, 17, 512]
    c = _random_case(len(lens), Hq, Hkv, D, [0] * len(lens), lens, seed=Hq * 7 + D, spike=True)
    ref = oo.decode_attention(c["q_decode"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"],
                              c["seq_lens"], c["scaling"], compute_dtype=torch.float32)
    for splits in (1, 2, 8):
        o = _run_decode(c, device, c["q_decode"], splits)
        # north-star bar for floating point: within 1e-3 of the fp32-accumulating torch-native oracle
        # (bf16 output rounding of |o|<~1 contributes up to 2^-9, hence atol 4e-3 on bf16 outputs)
        torch.testing.assert_close(o.float(), ref.float(), atol=4e-3, rtol=1e-2, msg=f"splits={splits}")


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (14, 2, 64), (16, 4, 128), (4, 4, 64)])
def test_extend_attention_random(device, Hq, Hkv, D):
    prefix = [0, 896, 5, 0, 63, 64, 200]
    extend = [130, 128, 1, 33, 65, 64, 7]
    c = _random_case(len(prefix), Hq, Hkv, D, prefix, extend, seed=Hq + D)
    ref = oo.extend_attention(c["q"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"],
                              c["seq_lens"], c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"],
                              compute_dtype=torch.float32)
    o = _run_extend(c, device)
    torch.testing.assert_close(o.float(), ref.float(), atol=4e-3, rtol=1e-2)


def test_extend_attention_noncausal(device):
    c = _random_case(3, 8, 2, 64, [0, 0, 0], [50, 64, 129], seed=9)
    ref = oo.extend_attention(c["q"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"],
                              c["seq_lens"], c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"],
                              causal=False, compute_dtype=torch.float32)
    o = _run_extend(c, device, causal=False)
    torch.testing.assert_close(o.float(), ref.float(), atol=4e-3, rtol=1e-2)


@pytest.mark.parametrize("shape", ["82", "42", "41"])
@pytest.mark.parametrize("Hq,Hkv,D,causal", [(32, 8, 128, True), (8, 1, 128, True), (14, 2, 64, True), (16, 4, 128, False),
                                             (4, 4, 64, False)])
def test_extend_attention_every_workgroup_shape(device, extend_shape, shape, Hq, Hkv, D, causal):
    """The launcher picks the workgroup shape from the grid size, so small test batches rarely reach the 8-wave
    kernel the bench's prefills run on: the `extend_shape` fixture forces each shape in turn (82 = 8 waves, 256 rows: the
    32x32 two-score-set kernel; 42 / 41 = the 4-wave forms) on a ragged batch with cached prefixes, one-token and
    tile-straddling extends, and a late dominating key (the deferred-rescale path)."""
    if shape == "41" and Hq // Hkv > 64:
        pytest.skip("the 64-row shape holds groups up to 64")
    extend_shape(shape)
    prefix = [0, 896, 5, 0, 63, 64, 200, 1000] if causal else [0, 0, 0, 0]
    extend = [130, 128, 1, 333, 65, 64, 7, 70] if causal else [50, 64, 129, 300]
    c = _random_case(len(prefix), Hq, Hkv, D, prefix, extend, seed=Hq + D + int(shape), spike=True)
    # the spike of _random_case follows the DECODE query of the last request: give the extend queries one too
    c["q"][-3] = (c["k_cache"][c["req_to_token"][c["req_pool_indices"][-1], 3].long(), 0] * 30).to(BF)
    ref = oo.extend_attention(c["q"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"],
                              c["seq_lens"], c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"],
                              causal=causal, compute_dtype=torch.float32)
    o = _run_extend(c, device, causal=causal)
    torch.testing.assert_close(o.float(), ref.float(), atol=4e-3, rtol=1e-2)


@pytest.mark.parametrize("Hq,Hkv,D,causal", [(32, 8, 128, True), (14, 2, 64, True), (16, 4, 128, False)])
def test_extend_attention_eight_wave_kernels_agree(device, extend_shape, Hq, Hkv, D, causal):
    """The three kernels an 8-wave bf16 launch can take -- the 32x32 two-score-set kernel (default), the ping-pong
    16x16x32 kernel it replaced (flag 2) and the general single-image kernel (flag 1) -- on the same launch shape: each
    within the oracle bar, and within 2^-7 of each other (a deferred row maximum scales P by up to 2^8 before its bf16
    rounding, so they are not bit-identical)."""
    prefix = [0, 896, 5, 0, 63, 64, 200, 1000] if causal else [0, 0, 0, 0]
    extend = [130, 128, 1, 333, 65, 64, 7, 70] if causal else [50, 64, 129, 300]
    c = _random_case(len(prefix), Hq, Hkv, D, prefix, extend, seed=Hq + D + 82, spike=True)
    c["q"][-3] = (c["k_cache"][c["req_to_token"][c["req_pool_indices"][-1], 3].long(), 0] * 30).to(BF)
    ref = oo.extend_attention(c["q"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"],
                              c["seq_lens"], c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"],
                              causal=causal, compute_dtype=torch.float32)
    outs = {}
    for name, flags in (("32x32", 0), ("ping-pong", 2), ("single-image", 1)):
        extend_shape("82", flags=flags)
        outs[name] = _run_extend(c, device, causal=causal)
        torch.testing.assert_close(outs[name].float(), ref.float(), atol=4e-3, rtol=1e-2, msg=name)
    for name in ("ping-pong", "single-image"):
        torch.testing.assert_close(outs["32x32"].float(), outs[name].float(), atol=2.0 ** -7, rtol=2.0 ** -7, msg=name)


@pytest.mark.parametrize("Hq,Hkv,D,causal", [(32, 8, 128, True), (8, 1, 128, True), (14, 2, 64, True), (16, 4, 128, False),
                                             (4, 4, 64, False), (4, 4, 128, True), (64, 1, 128, True)])
def test_extend_attention_32x32_form(device, extend_shape, Hq, Hkv, D, causal):
    """The 32x32-MFMA two-score-set kernel (one barrier per tile; the exponentials of tile t ride beside the score
    products of tile t + 1, its maxima beside the output products of tile t - 1; a raised maximum reaches O and l when
    everything exponentiated against the old one is inside them): ragged batch with cached prefixes, one-token and
    tile-straddling extends (walks of 1, 2, 3 and many tiles: each tail of the two-step loop), GQA groups 1 .. 64 (7:
    rows of the last tile stay empty), and dominating keys late in a walk (the deferred-rescale path)."""
    prefix = [0, 896, 5, 0, 63, 64, 200, 1000] if causal else [0, 0, 0, 0]
    extend = [130, 128, 1, 333, 65, 64, 7, 70] if causal else [50, 64, 129, 300]
    c = _random_case(len(prefix), Hq, Hkv, D, prefix, extend, seed=Hq + D + 3232, spike=True)
    c["q"][-3] = (c["k_cache"][c["req_to_token"][c["req_pool_indices"][-1], 3].long(), 0] * 30).to(BF)
    # ... and a key deep inside a long walk that dominates a query of the fourth request only from its tile on
    tok = c["req_to_token"][c["req_pool_indices"][3], 200].long()
    c["k_cache"][tok, 0] = (c["q"][int(sum(extend[:3])) + extend[3] - 20, 0] * 60).to(BF)
    ref = oo.extend_attention(c["q"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"],
                              c["seq_lens"], c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"],
                              causal=causal, compute_dtype=torch.float32)
    extend_shape("82")
    o = _run_extend(c, device, causal=causal)
    extend_shape("82", flags=2)
    o_pp = _run_extend(c, device, causal=causal)
    torch.testing.assert_close(o.float(), ref.float(), atol=4e-3, rtol=1e-2)
    torch.testing.assert_close(o.float(), o_pp.float(), atol=2.0 ** -7, rtol=2.0 ** -7)


def test_decode_equals_extend_of_one_token(device):
    """Size-independent property: decoding token n == extending by one token over prefix n-1."""
    lens = [700, 1024, 33]
    c = _random_case(len(lens), 32, 8, 128, [l - 1 for l in lens], [1, 1, 1], seed=77)
    o_ext = _run_extend(c, device)
    o_dec = _run_decode(c, device, c["q"], 2)
    torch.testing.assert_close(o_ext.float(), o_dec.float(), atol=4e-3, rtol=1e-2)


# ------------------------------------------------------------------ sampling
def test_argmax_and_softmax(device):
    torch.manual_seed(0)
    for V in (1000, 32000, 128256, 151936):
        logits = torch.randn((5, V)) * 3
        logits[2, 77] = logits[2].max() + 1
        logits[3, V - 1] = 50.0
        logits[4, 10] = logits[4, 20000 % V] = 60.0   # tie: first index wins
        ids = _k().argmax(logits.to(device)).cpu()
        assert torch.equal(ids, torch.argmax(logits, -1))
        ids16 = _k().argmax(logits.to(BF).to(device)).cpu()
        assert torch.equal(ids16, torch.argmax(logits.to(BF).float(), -1))
        temps = torch.tensor([1.0, 0.7, 1.3, 0.5, 2.0])
        p = _k().softmax_temperature_(logits.clone().to(device), temps.to(device)).cpu()
        ref = torch.softmax(logits / temps.view(-1, 1), dim=-1)
        torch.testing.assert_close(p, ref, atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("M,V,H", [(64, 128256, 4096), (1, 1000, 896), (17, 32000, 8192)])
def test_embedding_rmsnorm_equals_lookup_then_norm(device, M, V, H):
    """Round 6: the decode step's embedding lookup and the first layer's input norm in one launch (models/llama.py:433-437 then
    :349-353): the gathered rows bit-identical to F.embedding, the normed rows bit-identical to the RMSNorm kernel on them (the same
    arithmetic) and within the norm's bar of the oracle."""
    g = torch.Generator().manual_seed(M + H)
    table = (torch.randn((V, H), generator=g) * 0.05).to(BF).to(device)
    w = (1.0 + 0.1 * torch.randn(H, generator=g)).to(BF).to(device)
    ids = torch.randint(0, V, (M,), generator=g).to(device)
    hidden, normed = _k().embedding_rmsnorm(ids, table, w, 1e-5)
    want_h = torch.nn.functional.embedding(ids, table)
    assert torch.equal(hidden, want_h)
    assert torch.equal(normed, _k().rmsnorm(want_h, w, 1e-5))
    ref = oo.rmsnorm(want_h.cpu(), w.cpu(), 1e-5)
    assert float((normed.cpu().float() - ref.float()).abs().max()) <= 2.0 ** -7 * float(ref.float().abs().max())


@pytest.mark.parametrize("B,V", [(64, 128256), (3, 151936), (17, 32003), (200, 50257)])
def test_argmax_split_rows_semantics(device, B, V):
    """The column-range form (decode-sized batches of wide rows): first maximum on ties across ranges, NaN maximal,
    -0 == +0, all -inf rows, and the workspace re-arms itself (three calls through the same one)."""
    torch.manual_seed(B + V)
    logits = torch.randn((B, V)) * 2
    logits[0, 5] = logits[0, V - 3] = 40.0             # tie in the first and the last range: first index wins
    if B > 1:
        logits[1, V // 2 + 1] = float("nan")           # NaN beats everything
        logits[1, 7] = 1e30
    if B > 2:
        logits[2].fill_(-0.0)
        logits[2, V // 3] = 0.0                        # +0 after -0: equal, the first index (0) wins
    for dt in (torch.float32, BF):
        x = logits.to(dt)
        if dt == torch.float32 and (x.stride(0) * 4) % 16 != 0:
            continue                                   # unaligned rows take the one-workgroup-per-row kernel anyway
        want = torch.argmax(x.float(), -1)
        if B > 1:
            want[1] = V // 2 + 1
        for _ in range(3):
            got = _k().argmax(x.to(device)).cpu()
            assert torch.equal(got, want), (dt, (got != want).nonzero().flatten().tolist()[:5])
    ninf = torch.full((4, 65536), float("-inf"), dtype=BF)
    assert _k().argmax(ninf.to(device)).cpu().tolist() == [0, 0, 0, 0]
