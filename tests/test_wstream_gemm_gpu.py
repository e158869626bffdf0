"""GPU parity tests of the weight-streaming decode GEMM (wstream_gemm.hip) and its fused combine
epilogues against an fp64 reference on the same bf16 inputs (F.linear, srt/layers/linear.py:1596-1660)
and the oracle's silu_and_mul / fused_add_rmsnorm (activation.py:141-143, layernorm.py:786-820)."""
import pytest
import torch

from oracle import ops as oo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _k():
    from sglang_amd import kernels

    return kernels


def _ref_linear(x, w, bias=None):
    y = x.double() @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    return y


def _check(got, ref64, ulps=1.0):
    ref = ref64.float()
    err = (got.float() - ref).abs()
    tol = ref.abs() * 2.0 ** -8 * ulps + 1e-3
    assert bool((err <= tol).all()), f"max err {float(err.max())} (ref max {float(ref.abs().max())})"


@pytest.mark.parametrize("M", [1, 7, 16, 33, 48, 64, 65, 90, 128])
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (1024, 14336), (4864, 896), (112, 256), (16, 128)])
@pytest.mark.parametrize("nw,splits", [(None, None), (4, 1), (5, 2), (6, 1), (7, 2), (8, 3)])
def test_wstream_gemm_matches_fp64_reference(device, M, N, K, nw, splits):
    K_ = _k()
    if splits is not None and splits > K // 128:
        pytest.skip("more splits than K chunks")
    if M > 64 and nw is not None and nw > 5:
        pytest.skip("beyond 64 rows only 4- and 5-wave groups keep a 3-deep ring")
    g = torch.Generator().manual_seed(M * 131 + N + K)
    x = (torch.randn((M, K), generator=g) * 0.5).to(BF)
    w = (torch.randn((N, K), generator=g) * 0.05).to(BF)
    got = K_.wstream_gemm(x.to(device), w.to(device), waves_per_group=nw, splits=splits).cpu()
    _check(got, _ref_linear(x, w))


def test_wstream_gemm_bias_strides_and_untouched_padding(device):
    K_ = _k()
    g = torch.Generator().manual_seed(5)
    M, N, K = 19, 1152, 896
    xfull = torch.randn((M, K + 64), generator=g).to(BF).to(device)
    x = xfull[:, :K]
    w = (torch.randn((N, K), generator=g) * 0.05).to(BF).to(device)
    b = torch.randn(N, generator=g).to(BF).to(device)
    for splits in (1, 3):
        out_full = torch.zeros((M, N + 8), dtype=BF, device=device)
        K_.wstream_gemm(x, w, bias=b, out=out_full[:, :N], splits=splits)
        _check(out_full[:, :N].cpu(), _ref_linear(x.cpu(), w.cpu(), b.cpu()))
        assert float(out_full[:, N:].abs().max()) == 0.0


def test_wstream_gemm_split_k_is_deterministic_and_equals_one_split_order(device):
    K_ = _k()
    g = torch.Generator().manual_seed(9)
    x = torch.randn((64, 4096), generator=g).to(BF).to(device)
    w = (torch.randn((4096, 4096), generator=g) * 0.05).to(BF).to(device)
    outs = [K_.wstream_gemm(x, w, splits=8, waves_per_group=8).clone() for _ in range(5)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # another group size walks the K range from other staggered starting chunks: same products, another fp32
    # summation order, so only the last bf16 ulp of a few outputs may move
    alt = K_.wstream_gemm(x, w, splits=8, waves_per_group=4)
    d = (alt.float() - outs[0].float()).abs()
    assert float((d > 0).float().mean()) < 0.02
    assert bool((d <= outs[0].float().abs() * 2.0 ** -7 + 1e-3).all())
    _check(outs[0].cpu(), _ref_linear(x.cpu(), w.cpu()))


@pytest.mark.parametrize("M", [3, 64])
@pytest.mark.parametrize("splits", [1, 2, 4])
def test_wstream_silu_epilogue_equals_unfused_ops(device, M, splits):
    K_ = _k()
    g = torch.Generator().manual_seed(M)
    I, K = 1792, 1024
    x = torch.randn((M, K), generator=g).to(BF).to(device)
    w = (torch.randn((2 * I, K), generator=g) * 0.05).to(BF).to(device)
    gate_up = K_.wstream_gemm(x, w, splits=splits)
    want = oo.silu_and_mul(gate_up.cpu())
    got = K_.wstream_gemm(x, w, epilogue="silu_and_mul", splits=splits).cpu()
    d = (got.float() - want.float()).abs()
    # same products and rounding points; the one-pass form walks K from other starting chunks (fp32 summation
    # order) and expf may differ in the last place: a few gate / up values cross a bf16 rounding boundary (one ulp of a
    # gate around -5 moves silu(gate) by 2.5 %)
    assert float((d > 0).float().mean()) < 0.02
    assert bool((d <= want.float().abs() * 2.0 ** -4 + 1e-3).all())


@pytest.mark.parametrize("M", [1, 20, 64, 128])
@pytest.mark.parametrize("N,K,splits", [(4096, 4096, 8), (4096, 14336, 14), (896, 4864 - 4864 % 128, 1), (256, 512, 2)])
def test_wstream_add_rmsnorm_epilogue_equals_unfused_ops(device, M, N, K, splits):
    K_ = _k()
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn((M, K), generator=g).to(BF).to(device)
    w = (torch.randn((N, K), generator=g) * 0.03).to(BF).to(device)
    res = torch.randn((M, N), generator=g).to(BF).to(device)
    nw_ = (torch.rand(N, generator=g) + 0.5).to(BF).to(device)
    eps = 1e-5
    h = K_.wstream_gemm(x, w, splits=splits)
    want_y, want_res = oo.fused_add_rmsnorm(h.cpu(), res.cpu(), nw_.cpu(), eps)
    res2 = res.clone()
    got = K_.wstream_gemm(x, w, epilogue="add_rmsnorm", residual=res2, norm_weight=nw_, eps=eps, splits=splits)
    assert torch.equal(res2.cpu(), want_res), "residual stream must be bit-exact (same bf16 sum)"
    d = (got.cpu().float() - want_y.float()).abs()
    # same fp32 inputs; the row reduction order differs from torch's -> at most 1 bf16 ulp
    assert bool((d <= want_y.float().abs() * 2.0 ** -7 + 1e-6).all())
    assert float((d > 0).float().mean()) < 0.02


def test_wstream_rejects_unsupported_shapes(device):
    K_ = _k()
    x = torch.zeros((129, 256), dtype=BF, device=device)
    w = torch.zeros((64, 256), dtype=BF, device=device)
    with pytest.raises(RuntimeError):
        K_.wstream_gemm(x, w)
    with pytest.raises(RuntimeError):
        K_.wstream_gemm(x[:4, :200], w[:, :200])


@pytest.mark.parametrize("M", [1, 20, 50, 64])
@pytest.mark.parametrize("I,K,nw", [(14336, 4096, 7), (14336, 4096, None), (28672, 8192, 7), (1792, 1024, 4), (1792, 1024, 8), (4864, 896, 5), (24, 256, 6)])
def test_wstream_one_pass_silu_interleaved_tiles(device, M, I, K, nw):
    """The one-tile form of the fused silu epilogue (tiles_per_wave = 1: a wave's 16 weight rows are 8 gate rows + the 8 up rows
    of the same output columns, gate and up swapped across lane halves in the epilogue): Llama-3-8B / 70B gate_up in whole
    rounds of 256 workgroups x 7 waves, plus ragged tile counts, every waves-per-group the kernel has, a last workgroup with idle
    waves.  Same bars as the two-tile form, and the two forms agree with each other to the same bound."""
    K_ = _k()
    g = torch.Generator().manual_seed(M + I)
    x = torch.randn((M, K), generator=g).to(BF).to(device)
    w = (torch.randn((2 * I, K), generator=g) * 0.03).to(BF).to(device)
    gate_up = K_.wstream_gemm(x, w, splits=1, waves_per_group=4)
    want = oo.silu_and_mul(gate_up.cpu())
    if nw is None:                       # the form the policy would pick with the interleaved tiles enabled: 256 x 7 waves
        got = K_.wstream_gemm(x, w, epilogue="silu_and_mul", splits=1, waves_per_group=7, tiles_per_wave=1).cpu()
    else:
        got = K_.wstream_gemm(x, w, epilogue="silu_and_mul", splits=1, waves_per_group=nw, tiles_per_wave=1).cpu()
    d = (got.float() - want.float()).abs()
    assert float((d > 0).float().mean()) < 0.03
    assert bool((d <= want.float().abs() * 2.0 ** -4 + 1e-3).all())
    two = K_.wstream_gemm(x, w, epilogue="silu_and_mul", splits=1, waves_per_group=2 if I % 32 == 0 else None, tiles_per_wave=2).cpu() if I % 16 == 0 else None
    if two is not None:
        d2 = (got.float() - two.float()).abs()
        assert bool((d2 <= two.float().abs() * 2.0 ** -4 + 1e-3).all())
    # chunk-major output (what the fused decode layer hands to down_proj)
    if I % 128 == 0 and nw is not None:
        blk = K_.wstream_gemm(x, w, epilogue="silu_and_mul", splits=1, waves_per_group=nw, tiles_per_wave=1, out_blocked=True)
        assert torch.equal(K_.unblock(blk).cpu(), got)


@pytest.mark.parametrize("M", [1, 20, 64, 100, 128])
@pytest.mark.parametrize("I,K,nw", [(14336, 4096, None), (1792, 1024, 2), (1792, 1024, 3), (4864, 896, 4), (48, 256, None)])
def test_wstream_one_pass_silu_equals_unfused_ops(device, M, I, K, nw):
    """splits == 1: silu(gate) * up comes out of the GEMM's own epilogue (two tiles per wave)."""
    K_ = _k()
    g = torch.Generator().manual_seed(M + I)
    x = torch.randn((M, K), generator=g).to(BF).to(device)
    w = (torch.randn((2 * I, K), generator=g) * 0.03).to(BF).to(device)
    if M > 64 and nw is not None and nw > 2:
        pytest.skip("beyond 64 rows the one-pass form runs 2-wave groups")
    gate_up = K_.wstream_gemm(x, w, splits=1, waves_per_group=4)
    want = oo.silu_and_mul(gate_up.cpu())
    got = K_.wstream_gemm(x, w, epilogue="silu_and_mul", splits=1, waves_per_group=nw, tiles_per_wave=2).cpu()
    d = (got.float() - want.float()).abs()
    # same products and rounding points; the K walk starts at other staggered chunks (fp32 summation order), so a gate
    # or up value sitting on a bf16 rounding boundary may land on the other side.  One ulp of a gate around -5 moves
    # silu(gate) by 2.5 % (d ln silu / dg ~ 0.8 there, ulp 2^-5), hence the relative bound of 2^-4.
    assert float((d > 0).float().mean()) < 0.03
    assert bool((d <= want.float().abs() * 2.0 ** -4 + 1e-3).all())


@pytest.mark.parametrize("M", [1, 17, 64, 128])
@pytest.mark.parametrize("Hq,Hkv,D,K,bias,f32cache", [(32, 8, 128, 4096, False, False), (14, 2, 64, 896, True, False),
                                                      (8, 2, 64, 256, False, True), (4, 1, 128, 256, True, False)])
def test_wstream_qkv_rope_store_equals_unfused_ops(device, M, Hq, Hkv, D, K, bias, f32cache):
    """qkv GEMM + combine(rope, KV store) == linear -> rotary_embedding (forward_native) -> set_kv_buffer."""
    K_ = _k()
    g = torch.Generator().manual_seed(M * 7 + Hq)
    N = (Hq + 2 * Hkv) * D
    x = torch.randn((M, K), generator=g).to(BF).to(device)
    w = (torch.randn((N, K), generator=g) * 0.03).to(BF).to(device)
    b = torch.randn(N, generator=g).to(BF).to(device) if bias else None
    max_pos = 512
    cache = oo.cos_sin_cache(oo.rope_inv_freq(D, 10000.0), max_pos)
    cache = cache if f32cache else cache.to(BF)
    positions = torch.randint(0, max_pos, (M,), generator=g)
    slots = 200
    loc = torch.randperm(slots - 1, generator=g)[:M] + 1
    kc = torch.zeros((slots, Hkv, D), dtype=BF, device=device)
    vc = torch.zeros((slots, Hkv, D), dtype=BF, device=device)
    splits = 2 if K >= 512 else 1
    qkv = K_.wstream_gemm(x, w, bias=b, splits=splits).cpu()               # same split count: same accumulators
    q_ref, k_ref, v_ref = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    q_ref, k_ref = oo.rotary_embedding(positions, q_ref.clone(), k_ref.clone(), D, cache.cpu(), True)
    q = K_.wstream_qkv_rope(x, w, b, positions.to(device), cache.to(device), Hq, Hkv, D, kc, vc, loc.to(device), splits=splits)
    assert torch.equal(q.cpu(), q_ref.reshape(M, -1)), "rotated q must be bit-exact"
    assert torch.equal(kc.cpu()[loc], k_ref.reshape(M, Hkv, D)), "rotated k rows in the pool must be bit-exact"
    assert torch.equal(vc.cpu()[loc], v_ref.reshape(M, Hkv, D))
    untouched = torch.ones(slots, dtype=torch.bool)
    untouched[loc] = False
    assert float(kc.cpu()[untouched].abs().max()) == 0.0 and float(vc.cpu()[untouched].abs().max()) == 0.0


@pytest.mark.parametrize("M", [1, 23, 64])
@pytest.mark.parametrize("fp8,hnd,page", [(True, False, 1), (False, True, 16), (True, True, 8), (True, False, 4)])
@pytest.mark.parametrize("Hq,Hkv,D,K", [(32, 8, 128, 4096), (4, 2, 64, 256)])
def test_wstream_qkv_rope_store_in_the_pool_format(device, M, fp8, hnd, page, Hq, Hkv, D, K):
    """fp8 / HND pools: the fused combine writes the very bytes store_kv_cache(format) writes for the unfused
    rope output (set_kv_buffer, memory_pool.py:2364-2374 / 2061-2117), and nothing else in the pool."""
    K_ = _k()
    g = torch.Generator().manual_seed(M * 11 + Hq + page)
    N = (Hq + 2 * Hkv) * D
    x = torch.randn((M, K), generator=g).to(BF).to(device)
    w = (torch.randn((N, K), generator=g) * 0.03).to(BF).to(device)
    max_pos = 512
    cache = oo.cos_sin_cache(oo.rope_inv_freq(D, 10000.0), max_pos).to(BF)
    positions = torch.randint(0, max_pos, (M,), generator=g)
    slots = 256
    loc = (torch.randperm(slots - page, generator=g)[:M] + page).to(device)
    shape = (slots // page, Hkv, page, D) if hnd else (slots, Hkv, D)
    dt = torch.uint8 if fp8 else BF
    fmt = dict(kv_fp8=fp8, k_scale=0.75 if fp8 else 1.0, v_scale=1.5 if fp8 else 1.0, page_size=page, hnd=hnd)
    splits = 2 if K >= 512 else 1
    qkv = K_.wstream_gemm(x, w, splits=splits).cpu()
    q_ref, k_ref, v_ref = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    q_ref, k_ref = oo.rotary_embedding(positions, q_ref.clone(), k_ref.clone(), D, cache, True)
    kc_ref, vc_ref = torch.zeros(shape, dtype=dt, device=device), torch.zeros(shape, dtype=dt, device=device)
    K_.store_kv_cache(k_ref.reshape(M, -1).contiguous().to(device), v_ref.contiguous().to(device), kc_ref, vc_ref, loc,
                      num_kv_heads=Hkv, head_dim=D, **fmt)
    kc, vc = torch.zeros(shape, dtype=dt, device=device), torch.zeros(shape, dtype=dt, device=device)
    q = K_.wstream_qkv_rope(x, w, None, positions.to(device), cache.to(device), Hq, Hkv, D, kc, vc, loc, splits=splits, **fmt)
    assert torch.equal(q.cpu(), q_ref.reshape(M, -1))
    assert torch.equal(kc.cpu(), kc_ref.cpu()) and torch.equal(vc.cpu(), vc_ref.cpu())
    assert int((kc_ref.cpu().view(torch.uint8) != 0).sum()) > 0


def _blocked(x):
    """[M, K] -> chunk-major [K/128, M, 128]."""
    M, K = x.shape
    return x.view(M, K // 128, 128).permute(1, 0, 2).contiguous()


@pytest.mark.parametrize("M", [1, 20, 64, 100])
@pytest.mark.parametrize("N,K,splits", [(4096, 4096, None), (1024, 14336, 4), (256, 512, 1)])
def test_wstream_chunk_major_activations_change_no_bit(device, M, N, K, splits):
    """The chunk-major [K/128, M, 128] layout is an addressing choice: blocked in / blocked out give the very
    bits of the row-major call with the same decomposition."""
    K_ = _k()
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn((M, K), generator=g) * 0.5).to(BF).to(device)
    w = (torch.randn((N, K), generator=g) * 0.05).to(BF).to(device)
    nw, s = K_.choose_wstream_config(M, N, K)
    s = splits or s
    want = K_.wstream_gemm(x, w, waves_per_group=nw, splits=s)
    got_in = K_.wstream_gemm(_blocked(x), w, waves_per_group=nw, splits=s)
    got_out = K_.wstream_gemm(x, w, waves_per_group=nw, splits=s, out_blocked=True)
    assert got_out.shape == (N // 128, M, 128)
    assert torch.equal(got_in, want)
    assert torch.equal(K_.unblock(got_out), want)
    # a blocked activation with padding between its chunks (a view of a larger buffer)
    big = torch.zeros((K // 128, M + 3, 128), dtype=BF, device=device)
    big[:, :M] = _blocked(x)
    assert torch.equal(K_.wstream_gemm(big[:, :M], w, waves_per_group=nw, splits=s), want)


@pytest.mark.parametrize("M", [3, 64])
def test_wstream_chunk_major_epilogues(device, M):
    K_ = _k()
    g = torch.Generator().manual_seed(M)
    I, K, H = 1792, 1024, 1024
    x = torch.randn((M, K), generator=g).to(BF).to(device)
    w13 = (torch.randn((2 * I, K), generator=g) * 0.05).to(BF).to(device)
    for splits in (1, 2):
        want = K_.wstream_gemm(x, w13, epilogue="silu_and_mul", splits=splits)
        got = K_.wstream_gemm(_blocked(x), w13, epilogue="silu_and_mul", splits=splits, out_blocked=True)
        assert torch.equal(K_.unblock(got), want)
    w2 = (torch.randn((H, I), generator=g) * 0.05).to(BF).to(device)
    act = K_.wstream_gemm(x, w13, epilogue="silu_and_mul")
    nrm = torch.randn(H, generator=g).to(BF).to(device)
    res0 = torch.randn((M, H), generator=g).to(BF).to(device)
    r1, r2 = res0.clone(), res0.clone()
    want = K_.wstream_gemm(act, w2, epilogue="add_rmsnorm", residual=r1, norm_weight=nrm, eps=1e-5, splits=2)
    got = K_.wstream_gemm(_blocked(act), w2, epilogue="add_rmsnorm", residual=r2, norm_weight=nrm, eps=1e-5, splits=2,
                          out_blocked=True)
    assert torch.equal(K_.unblock(got), want) and torch.equal(r1, r2)


@pytest.mark.parametrize("M", [1, 16, 37, 64])
@pytest.mark.parametrize("N,K,nw,splits", [(128256, 4096, 4, 1), (4096, 4096, 2, 4), (6144, 896, 3, 2), (64, 256, 2, 1), (32, 128, 4, 1)])
def test_wstream_two_tiles_per_wave_plain_epilogues(device, M, N, K, nw, splits):
    """tiles_per_wave = 2 outside the silu form: a wave owns output tiles t and t + N/32 (the lm_head decomposition);
    bias, split-K partials, chunk-major output, against the fp64 reference and bit-for-bit against itself with a
    chunk-major input."""
    K_ = _k()
    if splits > K // 128:
        pytest.skip("more splits than K chunks")
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn((M, K), generator=g) * 0.5).to(BF).to(device)
    w = (torch.randn((N, K), generator=g) * 0.05).to(BF).to(device)
    b = torch.randn(N, generator=g).to(BF).to(device)
    ref = _ref_linear(x.cpu(), w.cpu(), b.cpu())
    got = K_.wstream_gemm(x, w, bias=b, waves_per_group=nw, splits=splits, tiles_per_wave=2)
    _check(got.cpu(), ref)
    if N % 128 == 0 and K % 128 == 0:
        blk = K_.wstream_gemm(_blocked(x), w, bias=b, waves_per_group=nw, splits=splits, tiles_per_wave=2, out_blocked=True)
        assert torch.equal(K_.unblock(blk), got)


def test_wstream_auto_decomposition_is_what_the_model_runs(device):
    """No explicit decomposition: choose_wstream_decomposition (two-tile waves for the wide single-split launches)."""
    K_ = _k()
    g = torch.Generator().manual_seed(3)
    for M, N, K in ((64, 128256, 4096), (5, 32000, 4096), (64, 28672, 4096)):
        x = (torch.randn((M, K), generator=g) * 0.5).to(BF).to(device)
        w = (torch.randn((N, K), generator=g) * 0.05).to(BF).to(device)
        assert K_.choose_wstream_decomposition(M, N, K)[1] == 2
        _check(K_.wstream_gemm(x, w).cpu(), _ref_linear(x.cpu(), w.cpu()))
