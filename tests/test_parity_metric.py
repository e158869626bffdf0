"""The error unit of the per-stage / per-layer parity bars (oracle/layer_parity.py `ulp_stats`): bf16 ulps of
max(|ref|, row rms).  Pinned on CPU so that a bar like "within one ulp" means what DESIGN.md section 4 says it means."""
import torch

from oracle.layer_parity import bf16_ulp, ulp_stats

BF = torch.bfloat16


def test_the_ulp_of_a_bf16_value_has_eight_significant_bits():
    x = torch.tensor([1.0, 1.5, 2.0, 3.99, 0.75, 100.0, 1e-3])
    exp = torch.floor(torch.log2(x))
    assert torch.equal(bf16_ulp(x), 2.0 ** (exp - 7))
    # next representable bf16 above x is x + ulp
    for v in (1.0, 2.0, 0.75, 100.0):
        up = torch.nextafter(torch.tensor(v, dtype=BF), torch.tensor(float("inf"), dtype=BF)).float()
        assert float(up - v) == float(bf16_ulp(torch.tensor(v)))


def test_ulp_stats_counts_in_units_of_the_reference_value_floored_at_the_row_rms():
    ref = torch.tensor([[4.0, 4.0, -4.0, 4.0]], dtype=BF)                             # row rms 4: every unit is ulp(4) = 2^-5
    got = ref.clone()
    s = ulp_stats(got, ref)
    assert s["frac_identical"] == 1.0 and s["max_ulp"] == 0.0 and s["mean_signed_ulp"] == 0.0
    inf, zero = torch.tensor(float("inf"), dtype=BF), torch.tensor(0.0, dtype=BF)
    got[0, 1] = torch.nextafter(ref[0, 1], inf)                                       # one step up from 4: +1 ulp
    got[0, 2] = torch.nextafter(torch.nextafter(ref[0, 2], zero), zero)               # two steps towards 0 from -4: they are
    s = ulp_stats(got, ref)                                                           # half-ulps of 4 below the binade: +1 ulp
    assert s["frac_identical"] == 0.5 and s["frac_within_1ulp"] == 1.0 and s["max_ulp"] == 1.0
    assert abs(s["mean_signed_ulp"] - 0.5) < 1e-6 and abs(s["rms_ulp"] - 0.5 ** 0.5) < 1e-6
    got[0, 0] = torch.nextafter(torch.nextafter(torch.nextafter(ref[0, 0], inf), inf), inf)   # three steps up: 3 ulp
    s = ulp_stats(got, ref)
    assert s["max_ulp"] == 3.0 and s["frac_within_2ulp"] == 0.75 and s["frac_within_1ulp"] == 0.75
    # values far below the row's rms are measured in ulps of the rms, not of themselves: noise around a zero crossing
    # does not count as thousands of ulps
    ref2 = torch.tensor([[4.0, -4.0, 4.0, 1e-4]], dtype=BF)
    got2 = ref2.clone()
    got2[0, 3] = -1e-4                                                                # a sign flip of a tiny value
    s2 = ulp_stats(got2, ref2)
    rms = float(ref2.float().pow(2).mean().sqrt())
    assert s2["max_ulp"] == abs(float(got2[0, 3].float() - ref2[0, 3].float())) / float(bf16_ulp(torch.tensor(rms)))
    assert s2["max_ulp"] < 0.02
