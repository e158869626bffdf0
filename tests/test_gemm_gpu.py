"""GPU parity tests of the weight-streaming skinny GEMM (dense decode projections) against an
fp32-accumulate reference on the same bf16 inputs (F.linear semantics, srt/layers/linear.py:1596-1660)."""
import pytest
import torch
import torch.nn.functional as F

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
    """|got - ref| <= ulps bf16 ulp of the reference magnitude (+ accumulation slack)."""
    ref = ref64.float()
    err = (got.float() - ref).abs()
    tol = ref.abs() * 2.0 ** -8 * ulps + 1e-3
    assert bool((err <= tol).all()), f"max err {float(err.max())} (ref max {float(ref.abs().max())})"


@pytest.mark.parametrize("M", [1, 7, 16, 33, 64])
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (1024, 14336), (4864, 896), (100, 256), (72, 40)])
@pytest.mark.parametrize("splits,ntw", [(1, 1), (1, 2), (None, None), (2, 2)])
def test_skinny_gemm_matches_fp32_reference(device, M, N, K, splits, ntw):
    K_ = _k()
    g = torch.Generator().manual_seed(M * 131 + N + K)
    x = (torch.randn((M, K), generator=g) * 0.5).to(BF)
    w = (torch.randn((N, K), generator=g) * 0.05).to(BF)
    if splits is not None and splits > (K + 127) // 128:
        pytest.skip("more splits than K chunks")
    got = K_.skinny_gemm(x.to(device), w.to(device), splits=splits, tiles_per_wave=ntw).cpu()
    _check(got, _ref_linear(x, w))


def test_skinny_gemm_bias_and_strides(device):
    K_ = _k()
    g = torch.Generator().manual_seed(5)
    M, N, K = 19, 1152, 896
    xfull = (torch.randn((M, K + 64), generator=g)).to(BF).to(device)
    x = xfull[:, :K]                                     # row stride K + 64
    w = (torch.randn((N, K), generator=g) * 0.05).to(BF).to(device)
    b = torch.randn(N, generator=g).to(BF).to(device)
    out_full = torch.zeros((M, N + 8), dtype=BF, device=device)
    K_.skinny_gemm(x, w, bias=b, out=out_full[:, :N], splits=2)
    _check(out_full[:, :N].cpu(), _ref_linear(x.cpu(), w.cpu(), b.cpu()))
    assert float(out_full[:, N:].abs().max()) == 0.0


def test_skinny_gemm_split_k_is_deterministic(device):
    K_ = _k()
    g = torch.Generator().manual_seed(9)
    x = torch.randn((64, 4096), generator=g).to(BF).to(device)
    w = (torch.randn((4096, 4096), generator=g) * 0.05).to(BF).to(device)
    outs = [K_.skinny_gemm(x, w, splits=8).clone() for _ in range(5)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    _check(outs[0].cpu(), _ref_linear(x.cpu(), w.cpu()))


@pytest.mark.parametrize("M", [3, 64])
def test_skinny_gemm_fused_silu_equals_unfused_torch_ops(device, M):
    """fuse_silu reproduces linear -> SiluAndMul.forward_native (activation.py:141-143) rounding."""
    K_ = _k()
    g = torch.Generator().manual_seed(M)
    N, K = 1792, 1024
    x = torch.randn((M, K), generator=g).to(BF).to(device)
    w = (torch.randn((2 * N, K), generator=g) * 0.05).to(BF).to(device)
    gate_up = K_.skinny_gemm(x, w, splits=1)
    want = oo.silu_and_mul(gate_up.cpu())
    got = K_.skinny_gemm(x, w, fuse_silu=True, splits=1).cpu()
    # identical accumulators and rounding points: only the silu's exp may differ in the last bf16 ulp
    d = (got.float() - want.float()).abs()
    assert float((d > 0).float().mean()) < 0.005
    assert bool((d <= want.float().abs() * 2.0 ** -7 + 1e-6).all())
    # split-K changes the accumulation order: gate and up may each move by one bf16 ulp
    got2 = K_.skinny_gemm(x, w, fuse_silu=True, splits=4).cpu()
    d2 = (got2.float() - want.float()).abs()
    assert bool((d2 <= want.float().abs() * 2.0 ** -5 + 1e-3).all())
