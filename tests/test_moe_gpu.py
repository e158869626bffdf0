"""GPU parity tests of the MoE path (router, align, grouped GEMM, sum-reduce, fused layer) against the
CPU oracle (oracle/ops.py: fused_topk / moe_forward, restating srt/layers/moe/topk.py:690-736 and
fused_moe_native.py:61-164), the golden fixture from the real reference (tests/golden/moe_native.pt)
and the reference's own integer spec for moe_align_block_size."""
import pytest
import torch

from oracle import ops as oo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _k():
    from sglang_amd import kernels

    return kernels


def _load(golden_dir, name):
    return torch.load(golden_dir / name, weights_only=False)


def _align_spec(topk_ids: torch.Tensor, num_experts: int, block_size: int):
    """Stable counting sort spec (test/registered/kernels/ops/moe/test_moe_align_block_size.py:23-140)."""
    flat = topk_ids.flatten().tolist()
    numel = len(flat)
    sorted_ids, expert_ids = [], []
    for e in range(-1, num_experts):
        idx = [i for i, v in enumerate(flat) if v == e]
        if not idx:
            continue
        pad = (-len(idx)) % block_size
        sorted_ids += idx + [numel] * pad
        expert_ids += [e] * ((len(idx) + pad) // block_size)
    return sorted_ids, expert_ids, len(sorted_ids)


@pytest.mark.parametrize("block_size,num_tokens,topk,num_experts", [(32, 1, 1, 64), (128, 48, 1, 128), (64, 103, 4, 256),
                                                                     (16, 64, 2, 8), (64, 4096, 2, 8), (32, 7, 2, 8),
                                                                     (16, 3000, 8, 260)])
def test_moe_align_block_size_matches_spec(device, block_size, num_tokens, topk, num_experts):
    K = _k()
    g = torch.Generator().manual_seed(num_tokens)
    ids = torch.argsort(torch.rand((num_tokens, num_experts), generator=g), dim=1)[:, :topk].to(torch.int32)
    s, e, post = K.moe_align_block_size(ids.to(device), block_size, num_experts)
    ws, we, wp = _align_spec(ids, num_experts, block_size)
    assert int(post) == wp
    assert s.cpu()[:wp].tolist() == ws                       # stable order, pads == numel
    assert e.cpu()[: wp // block_size].tolist() == we
    assert (e.cpu()[wp // block_size:] == -1).all()


def test_moe_align_filtered_experts(device):
    K = _k()
    ids = torch.tensor([[0, -1], [2, 0], [-1, -1], [1, 2]], dtype=torch.int64)
    s, e, post = K.moe_align_block_size(ids.to(device), 4, 3)
    ws, we, wp = _align_spec(ids, 3, 4)
    assert int(post) == wp and s.cpu()[:wp].tolist() == ws and e.cpu()[: wp // 4].tolist() == we
    assert we[0] == -1


@pytest.mark.parametrize("M,E,k", [(9, 8, 2), (1, 8, 2), (300, 64, 6), (17, 256, 8), (5, 3, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_topk_softmax_matches_oracle(device, M, E, k, dtype):
    K = _k()
    g = torch.Generator().manual_seed(M * E + k)
    logits = (torch.randn((M, E), generator=g) * 2).to(dtype)
    for renorm in (True, False):
        w, ids = K.topk_softmax(logits.to(device), k, renorm)
        rw, rids = oo.fused_topk(logits, k, renorm)
        torch.testing.assert_close(w.cpu(), rw, rtol=2e-6, atol=1e-8)
        # torch.topk orders tied scores arbitrarily (bf16 logits tie often); the kernel takes the
        # lowest expert id first like the reference's CUDA kernel.  Ids must agree wherever the
        # scores are distinct, and always select the same scores.
        probs = logits.float().softmax(dim=-1)
        assert torch.equal(probs.gather(1, ids.cpu().long()), probs.gather(1, rids.long()))
        srt = probs.sort(dim=-1, descending=True)[0]
        no_ties = (srt[:, :-1] != srt[:, 1:]).all(dim=1)
        assert torch.equal(ids.cpu()[no_ties], rids[no_ties])


def test_topk_softmax_golden_and_degenerate(device, golden_dir):
    """Golden ids/weights from the real reference; all-equal logits pick the lowest expert ids
    (test/registered/moe/test_topk_renormalize_degenerate.py)."""
    K = _k()
    c = _load(golden_dir, "moe_native.pt")
    w, ids = K.topk_softmax(c["router_logits"].to(device), c["topk"], True)
    assert torch.equal(ids.cpu().long(), c["topk_ids"].long())
    torch.testing.assert_close(w.cpu(), c["topk_weights"], rtol=2e-6, atol=1e-8)
    w, ids = K.topk_softmax(torch.zeros((3, 8), device=device), 2, True)
    assert ids.cpu().tolist() == [[0, 1]] * 3
    torch.testing.assert_close(w.cpu(), torch.full((3, 2), 0.5))


def test_moe_sum_reduce(device):
    K = _k()
    g = torch.Generator().manual_seed(0)
    x = torch.randn((37, 2, 4096), generator=g)
    out = torch.empty((37, 4096), dtype=BF, device=device)
    K.moe_sum_reduce(x.to(device), out, 1.0)
    assert torch.equal(out.cpu(), x.sum(dim=1).to(BF))
    xb = x.to(BF)
    K.moe_sum_reduce(xb.to(device), out, 2.5)
    assert torch.equal(out.cpu(), (xb.float().sum(dim=1) * 2.5).to(BF))


def _moe_check(got: torch.Tensor, want: torch.Tensor):
    """bf16 outputs of the same arithmetic with a different GEMM accumulation order."""
    d = (got.float() - want.float()).abs()
    tol = want.float().abs() * 2.0 ** -6 + 2e-3 * float(want.float().abs().max())
    assert bool((d <= tol).all()), f"max err {float(d.max())} at scale {float(want.float().abs().max())}"


def test_fused_experts_golden(device, golden_dir):
    """tests/golden/moe_native.pt: the real reference's fused_moe_forward_native (einsum form) output."""
    K = _k()
    c = _load(golden_dir, "moe_native.pt")
    out = K.fused_experts(c["x"].to(device), c["w13"].to(device), c["w2"].to(device), c["topk_weights"].to(device),
                          c["topk_ids"].to(torch.int32).to(device))
    _moe_check(out.cpu(), c["out_einsum"])
    want = oo.moe_forward(c["x"], c["w13"], c["w2"], c["topk_weights"], c["topk_ids"])
    _moe_check(out.cpu(), want)


@pytest.mark.parametrize("M,E,k,N,Kd", [(1, 8, 2, 512, 256), (64, 8, 2, 1792, 1024), (200, 8, 2, 256, 512),
                                        (33, 16, 4, 384, 128)])
def test_fused_experts_matches_oracle(device, M, E, k, N, Kd):
    K = _k()
    g = torch.Generator().manual_seed(M + E)
    x = torch.randn((M, Kd), generator=g).to(BF)
    w13 = (torch.randn((E, 2 * N, Kd), generator=g) * 0.05).to(BF)
    w2 = (torch.randn((E, Kd, N), generator=g) * 0.05).to(BF)
    logits = torch.randn((M, E), generator=g)
    tw, ti = oo.fused_topk(logits, k, True)
    out = K.fused_experts(x.to(device), w13.to(device), w2.to(device), tw.to(device), ti.to(device))
    want = oo.moe_forward(x, w13, w2, tw, ti)
    _moe_check(out.cpu(), want)


def test_grouped_gemm_row_gather_and_scale(device):
    """The grouped GEMM alone vs a per-pair loop (invoke_fused_moe_kernel semantics,
    fused_moe_triton_kernels.py:324-770): gather a[id // topk], scatter to c[id], router weight."""
    K = _k()
    g = torch.Generator().manual_seed(4)
    M, E, k, N, Kd, bm = 50, 8, 2, 320, 384, 32
    a = torch.randn((M, Kd), generator=g).to(BF)
    w = (torch.randn((E, N, Kd), generator=g) * 0.05).to(BF)
    ids = torch.argsort(torch.rand((M, E), generator=g), dim=1)[:, :k].to(torch.int32)
    tw = torch.rand((M, k), generator=g)
    s, e, post = K.moe_align_block_size(ids.to(device), bm, E)
    for splits in (1, 3):
        c = torch.zeros((M * k, N), dtype=BF, device=device)
        K.moe_grouped_gemm(a.to(device), w.to(device), c, s, e, post, tw.reshape(-1).to(device), True, k, M * k, bm,
                           splits=splits)
        want = torch.empty((M * k, N))
        for i in range(M * k):
            want[i] = (a[i // k].double() @ w[int(ids.flatten()[i])].double().t()).float() * tw.flatten()[i]
        d = (c.cpu().float() - want).abs()
        assert bool((d <= want.abs() * 2.0 ** -7 + 1e-3).all())


def test_mixtral_block_module(device):
    from sglang_amd.harness.models import CONFIGS
    from sglang_amd.harness.moe_block import SparseMoeBlock

    cfg = CONFIGS["tiny-mixtral"]
    blk = SparseMoeBlock(cfg, "model.layers.0.block_sparse_moe", "cpu", device, 0, 1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn((21, cfg.hidden_size), generator=g).to(BF)
    got = blk(x.to(device)).cpu()
    logits = torch.nn.functional.linear(x, blk.gate.weight.data.cpu())
    tw, ti = oo.fused_topk(logits, cfg.num_experts_per_tok, True)
    want = oo.moe_forward(x, blk.experts.w13_weight.data.cpu(), blk.experts.w2_weight.data.cpu(), tw, ti)
    _moe_check(got, want)


@pytest.mark.parametrize("M,E,k,N,Kd,bm", [(50, 8, 2, 320, 384, 32), (64, 8, 2, 1792, 1024, 16), (3, 4, 2, 64, 128, 64),
                                           (130, 8, 2, 256, 256, 48)])
def test_wstream_moe_gemm_row_gather_scale_and_silu(device, M, E, k, N, Kd, bm):
    """The weight-streaming grouped form vs a per-pair loop (invoke_fused_moe_kernel semantics,
    fused_moe_triton_kernels.py:324-770): gather a[id // topk], scatter to c[id], router weight, fp32 / bf16
    outputs, and the silu_and_mul epilogue against the unfused ops on the same accumulators."""
    K = _k()
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn((M, Kd), generator=g).to(BF)
    w = (torch.randn((E, N, Kd), generator=g) * 0.05).to(BF)
    ids = torch.argsort(torch.rand((M, E), generator=g), dim=1)[:, :k].to(torch.int32)
    tw = torch.rand((M, k), generator=g)
    s, e, post = K.moe_align_block_size(ids.to(device), bm, E)
    want = torch.empty((M * k, N))
    for i in range(M * k):
        want[i] = (a[i // k].double() @ w[int(ids.flatten()[i])].double().t()).float()
    for f32 in (False, True):
        c = torch.zeros((M * k, N), dtype=torch.float32 if f32 else BF, device=device)
        K.moe_wstream_gemm(a.to(device), w.to(device), c, s, e, post, tw.reshape(-1).to(device), True, k, M * k, bm,
                           round_before_scale=f32)
        ref = (want.to(BF).float() if f32 else want) * tw.flatten()[:, None]
        d = (c.cpu().float() - ref).abs()
        assert bool((d <= ref.abs() * 2.0 ** -7 + 1e-3).all()), float(d.max())
    # silu form: w = [gate rows | up rows] of N/2 columns each
    c_plain = torch.zeros((M * k, N), dtype=BF, device=device)
    K.moe_wstream_gemm(a.to(device), w.to(device), c_plain, s, e, post, None, False, k, M * k, bm)
    c_silu = torch.zeros((M * k, N // 2), dtype=BF, device=device)
    K.moe_wstream_gemm(a.to(device), w.to(device), c_silu, s, e, post, None, False, k, M * k, bm, fuse_silu=True)
    # Same products and rounding points; the two launches walk K from other staggered starting chunks (fp32 summation
    # order), so a gate / up accumulator that sits on a bf16 rounding boundary may round the other way in the fused
    # launch.  That is the ONLY licence: every fused output must be silu_and_mul of bf16 (gate, up) values at most one
    # ulp away from the plain launch's, to one ulp of the result (the kernel's expf against torch's silu) -- an epilogue or ordering bug (percent-level errors on
    # arbitrary elements) cannot pass -- and all but a few elements must come from the unshifted pair.
    plain = c_plain.cpu()
    gate, up = plain[:, : N // 2], plain[:, N // 2:]

    def neighbour(x, step):                      # the bf16 value `step` ulps further from zero (sign-magnitude bits)
        return (x.contiguous().view(torch.int16) + step).view(BF)

    def ulp(x):                                  # one bf16 ulp at magnitude x (8 significant bits)
        return torch.ldexp(torch.ones_like(x), torch.frexp(x)[1] - 8)

    got = c_silu.cpu().float()
    best = torch.full_like(got, float("inf"))
    centre = None
    for dg in (0, -1, 1):
        for du in (0, -1, 1):
            cand = oo.silu_and_mul(torch.cat([neighbour(gate, dg), neighbour(up, du)], dim=1)).float()
            if centre is None:
                centre = cand
            best = torch.minimum(best, (got - cand).abs() - ulp(torch.maximum(got.abs(), cand.abs())))
    # (bf16 differences and ulps are exact in fp32.)  One more licence, for operands next to zero only: where |gate| or
    # |up| < 2^-8 the fp32 summation-order noise of a K = 1024 dot product (~2^-16 absolute) exceeds the value's own bf16
    # ulp -- cancellation -- so there the bound is the first-order propagation of that absolute noise,
    # |d silu/dg| |up| dg + |silu(gate)| du, plus two output ulps.
    gf, uf = gate.float(), up.float()
    eta = 2.0 ** -16
    small = (gf.abs() < 2.0 ** -8) | (uf.abs() < 2.0 ** -8)
    sg = torch.sigmoid(gf)
    slope = (sg * (1 + gf * (1 - sg))).abs()
    bound = 1.05 * (slope * uf.abs() * torch.maximum(ulp(gf), torch.tensor(eta)) + (gf * sg).abs() * torch.maximum(ulp(uf), torch.tensor(eta))) \
        + 2 * ulp(torch.maximum(got.abs(), centre.abs()))
    near_zero_ok = small & ((got - centre).abs() <= bound)
    bad = (best > 0) & ~near_zero_ok
    assert not bool(bad.any()), (int(bad.sum()), float(best[bad].max()))
    assert float(((best > 0) & near_zero_ok).float().mean()) < 1e-3          # the near-zero licence is rarely needed
    assert float(((got - centre).abs() > ulp(torch.maximum(got.abs(), centre.abs()))).float().mean()) < 0.02


@pytest.mark.parametrize("M,E,k,N,Kd", [(900, 8, 2, 512, 1024), (2048, 8, 2, 7168, 4096), (1500, 4, 1, 320, 192)])
def test_fused_experts_prefill_sized_batches_take_the_tiled_form(device, M, E, k, N, Kd):
    """>= 96 rows per expert: the row-tiled MFMA grouped GEMM (moe_tiled_gemm.hip, 128-row tiles) with the
    silu_and_mul epilogue and the router-weighted fp32 down projection; (2048, 8, 2, 7168, 4096) is one TP=2 rank of
    Mixtral-8x7B at a prefill batch.  The reference arithmetic (fused_moe_native.py:61-164) runs on the GPU here."""
    K = _k()
    assert M * k // E >= K.MOE_TILED_MIN_ROWS_PER_EXPERT
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn((M, Kd), generator=g).to(BF).to(device)
    w13 = (torch.randn((E, 2 * N, Kd), generator=g) * 0.03).to(BF).to(device)
    w2 = (torch.randn((E, Kd, N), generator=g) * 0.03).to(BF).to(device)
    logits = torch.randn((M, E), generator=g).to(device)
    tw, ti = oo.fused_topk(logits, k, True)
    out = K.fused_experts(x, w13, w2, tw, ti)
    want = oo.moe_forward(x, w13, w2, tw, ti)
    _moe_check(out.cpu(), want.cpu())


@pytest.mark.parametrize("plan", [(256, 256, 256), (256, 256, 128), (256, 128, 128)])
@pytest.mark.parametrize("M,E,k,N,Kd", [(4096, 8, 2, 7168, 4096), (3000, 4, 2, 448, 320), (1100, 2, 1, 1024, 1024)])
def test_fused_experts_256_row_tiles(device, monkeypatch, plan, M, E, k, N, Kd):
    """The 256 x 256 x 64 form (moe_gemm256_kernel: 8 waves, two LDS-DMA stages, XCD-patched order) for the up projection
    (silu epilogue), the down projection (router weight, fp32), or -- over the same 256-row alignment -- the 128-row tiles;
    (4096, 8, 2, 7168, 4096) is one TP=2 rank of Mixtral-8x7B, the others have ragged column tiles, K = 5 steps and a
    last row block that is mostly padding.  Same bars as the 128-row form, and the forms agree to one ulp."""
    K = _k()
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn((M, Kd), generator=g).to(BF).to(device)
    w13 = (torch.randn((E, 2 * N, Kd), generator=g) * 0.03).to(BF).to(device)
    w2 = (torch.randn((E, Kd, N), generator=g) * 0.03).to(BF).to(device)
    logits = torch.randn((M, E), generator=g).to(device)
    tw, ti = oo.fused_topk(logits, k, True)
    monkeypatch.setattr(K, "MOE_TILE_OVERRIDE", plan)
    out = K.fused_experts(x, w13, w2, tw, ti)
    want = oo.moe_forward(x, w13, w2, tw, ti)
    _moe_check(out.cpu(), want.cpu())
    monkeypatch.setattr(K, "MOE_TILE_OVERRIDE", (128, 128, 128))
    base = K.fused_experts(x, w13, w2, tw, ti)
    d = (out.float() - base.float()).abs()
    assert float((d > 2.0 ** -7 * base.float().abs() + 1e-3).float().mean()) < 1e-3, float(d.max())


def test_moe_tile_plan_policy():
    K = _k()
    assert K.moe_tile_plan(2 * 512, 8, 7168, 4096) == (128, 128, 128)                  # 128 rows per expert: the 128-row form
    assert K.moe_tile_plan(2 * 2048, 8, 14336, 4096) == (128, 128, 128)                 # 512 rows: padding to 256 costs too much
    assert K.moe_tile_plan(2 * 4096, 8, 14336, 4096) == (256, 256, 256)                 # Mixtral prefill: 1024 rows per expert


def test_tiled_grouped_gemm_row_gather_and_scale(device):
    """The tiled grouped GEMM alone vs a per-pair loop: gather a[id // topk], scatter to c[id], router weight, ragged
    expert loads incl. an expert with no rows, N not a multiple of the tile."""
    K = _k()
    g = torch.Generator().manual_seed(8)
    M, E, k, N, Kd = 700, 6, 2, 200, 256
    bm = K.moe_tiled_gemm_block_m()
    a = torch.randn((M, Kd), generator=g).to(BF)
    w = (torch.randn((E, N, Kd), generator=g) * 0.05).to(BF)
    ids = torch.argsort(torch.rand((M, E - 1), generator=g), dim=1)[:, :k].to(torch.int32)      # expert E-1 gets nothing
    tw = torch.rand((M, k), generator=g)
    s, e, post = K.moe_align_block_size(ids.to(device), bm, E)
    for out_dtype in (BF, torch.float32):
        c = torch.zeros((M * k, N), dtype=out_dtype, device=device)
        K.moe_tiled_gemm(a.to(device), w.to(device), c, s, e, post, tw.reshape(-1).to(device), True, k, M * k, bm)
        want = (torch.einsum("ik,ink->in", a.double()[torch.arange(M * k) // k], w.double()[ids.flatten().long()]).float()
                * tw.flatten()[:, None])
        d = (c.cpu().float() - want).abs()
        assert bool((d <= want.abs() * 2.0 ** -7 + 1e-3).all()), float(d.max())
