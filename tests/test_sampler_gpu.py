"""GPU parity tests of the gfx950 sampler (top-k / top-p / min-p + deterministic gumbel)
against the CPU oracle (oracle/host.py, restating srt/layers/sampler.py:567-750) and the
golden fixture generated from the real reference (tests/golden/sampler_torch.pt)."""
import numpy as np
import pytest
import torch

from oracle import host as oh

pytestmark = pytest.mark.gpu

TOP_K_ALL = 1 << 30


def _k():
    from sglang_amd import kernels

    return kernels


def _load(golden_dir, name):
    return torch.load(golden_dir / name, weights_only=False)


def _kept_counts(kept_sorted: torch.Tensor) -> torch.Tensor:
    return (kept_sorted > 0).sum(dim=1).to(torch.int32)


def test_murmur_known_answer_through_sampler(device):
    """test/registered/sampling/test_deterministic_gumbel_u1.py:14-21: hash(seed, 7371, 248146) = 0xFFFFFFFF,
    i.e. u = 1 -> the clamped gumbel is the largest possible, so with uniform probs over the
    248320-token vocabulary the sampler must return column 248146."""
    K = _k()
    V = 248320
    seed, pos, col = 6469398791980356130, 7371, 248146
    assert int(oh.murmur_hash32(np.array([seed], dtype=np.uint64), np.array([pos]), np.array([col]))[0, 0]) == 0xFFFFFFFF
    probs = torch.full((1, V), 1.0 / V, device=device)
    ids = K.top_k_top_p_min_p_sample(probs, None, None, None, torch.tensor([seed], device=device),
                                     torch.tensor([pos], device=device), filtered=False)
    ref = oh.sampling_from_probs(probs.cpu(), torch.tensor([seed]), torch.tensor([pos]))
    assert ids.cpu().tolist() == ref.tolist() == [col]


def test_kept_set_matches_reference_golden(device, golden_dir):
    """n_keep of every row equals the real reference's kept count (with and without min_p)."""
    K = _k()
    c = _load(golden_dir, "sampler_torch.pt")
    probs = c["probs"].to(device)
    B = probs.shape[0]
    seeds = torch.arange(B, device=device) + 1
    pos = torch.zeros(B, dtype=torch.int64, device=device)
    for need_min_p in (False, True):
        ids, n_keep = K.top_k_top_p_min_p_sample(probs, c["top_ks"].to(device), c["top_ps"].to(device),
                                                 c["min_ps"].to(device) if need_min_p else None, seeds, pos,
                                                 return_n_keep=True)
        assert n_keep.cpu().tolist() == _kept_counts(c[f"kept_sorted_minp{int(need_min_p)}"]).tolist()
        # every sampled token lies inside the reference's kept set
        kept = c[f"kept_sorted_minp{int(need_min_p)}"]
        order = c["probs"].sort(dim=-1, descending=True)[1]
        for b in range(B):
            allowed = set(order[b, : int((kept[b] > 0).sum())].tolist())
            assert int(ids[b]) in allowed


@pytest.mark.parametrize("V", [1000, 32000, 128256])
def test_seeded_sampling_matches_oracle(device, V):
    """Bit-exact token ids vs the torch path for seeded top-k/top-p sampling (small nuclei, LDS ranking)."""
    K = _k()
    g = torch.Generator().manual_seed(V)
    B = 16
    logits = torch.randn((B, V), generator=g) * 4
    probs = torch.softmax(logits, dim=-1)
    top_ks = torch.tensor([50, 1, 20, 5, 1000, 2, 64, 100] * 2, dtype=torch.int32)
    top_ps = torch.tensor([0.9, 1.0, 0.5, 0.3, 0.8, 0.95, 0.7, 0.6] * 2)
    seeds = torch.randint(0, 2 ** 62, (B,), generator=g)
    pos = torch.randint(0, 4096, (B,), generator=g)
    ref, kept, _ = oh.top_k_top_p_min_p_sampling_from_probs(probs.clone(), top_ks, top_ps, None, False, seeds, pos,
                                                            return_kept=True)
    ids, n_keep = K.top_k_top_p_min_p_sample(probs.to(device), top_ks.to(device), top_ps.to(device), None,
                                             seeds.to(device), pos.to(device), return_n_keep=True)
    assert n_keep.cpu().tolist() == _kept_counts(kept).tolist()
    assert ids.cpu().tolist() == ref.tolist()


def test_large_nucleus_uses_global_radix_sort(device):
    """top_p close to 1 on a flat distribution keeps tens of thousands of tokens: the ranking goes
    through the global-memory radix sort and must still match the torch sort order bit-exactly."""
    K = _k()
    g = torch.Generator().manual_seed(11)
    B, V = 4, 50000
    probs = torch.softmax(torch.randn((B, V), generator=g), dim=-1)
    top_ks = torch.tensor([TOP_K_ALL, 30000, TOP_K_ALL, 5000], dtype=torch.int32)
    top_ps = torch.tensor([0.9, 1.0, 0.5, 0.99])
    seeds = torch.randint(0, 2 ** 62, (B,), generator=g)
    pos = torch.randint(0, 4096, (B,), generator=g)
    ref, kept, _ = oh.top_k_top_p_min_p_sampling_from_probs(probs.clone(), top_ks, top_ps, None, False, seeds, pos,
                                                            return_kept=True)
    ids, n_keep = K.top_k_top_p_min_p_sample(probs.to(device), top_ks.to(device), top_ps.to(device), None,
                                             seeds.to(device), pos.to(device), return_n_keep=True)
    want = _kept_counts(kept)
    assert all(int(n) > K.sampling_lds_keep() for n in want)
    # the top-p rule is the reference's own arithmetic on the sorted list (prefix sums in double rounded to fp32 as
    # torch.cumsum does on the CPU, the fp32 subtraction and comparison of sampler.py:577-580): same cut, same ids
    assert n_keep.cpu().tolist() == want.tolist()
    assert ids.cpu().tolist() == ref.tolist()
    # without the workspace such rows are refused, not silently mis-sampled
    ids2 = K.top_k_top_p_min_p_sample(probs.to(device), top_ks.to(device), top_ps.to(device), None,
                                      seeds.to(device), pos.to(device), use_workspace=False)
    assert ids2.cpu().tolist() == [-1] * B


@pytest.mark.parametrize("V,scale", [(128256, 1.0), (128256, 3.0), (50000, 0.3), (32000, 6.0)])
def test_top_p_cut_is_the_references_own_arithmetic(device, V, scale):
    """The cut of sampler.py:577-580 bit for bit, where it is hardest: top_p close to 1 puts the cut into the flat far tail
    (p ~ 1e-7 .. 1e-9, well below one fp32 ulp of the prefix sum, so dozens of ranks sit inside the rounding noise of
    the reference's cumsum and its zeroing is not even monotone in the rank), top_p = 1.0 exactly (a softmax row's fp32
    values sum to 1 + a few 1e-7: the reference DOES zero part of the tail), small nuclei in LDS and large ones through
    the global radix sort, with and without a top-k cut.  Kept counts and token ids must equal the oracle's."""
    K = _k()
    g = torch.Generator().manual_seed(V + int(scale * 10))
    B = 12
    probs = torch.softmax(torch.randn((B, V), generator=g) * scale, dim=-1)
    top_ks = torch.tensor([TOP_K_ALL, TOP_K_ALL, TOP_K_ALL, TOP_K_ALL, 40000, TOP_K_ALL, 1500, TOP_K_ALL, 50, TOP_K_ALL, 3000,
                           TOP_K_ALL], dtype=torch.int32)
    top_ps = torch.tensor([0.999, 1.0, 0.9999, 0.99, 1.0, 0.99999, 0.999, 0.95, 1.0, 0.999999, 0.9999, 0.5])
    seeds = torch.randint(0, 2 ** 62, (B,), generator=g)
    pos = torch.randint(0, 4096, (B,), generator=g)
    ref, kept, _ = oh.top_k_top_p_min_p_sampling_from_probs(probs.clone(), top_ks, top_ps, None, False, seeds, pos,
                                                            return_kept=True)
    ids, n_keep = K.top_k_top_p_min_p_sample(probs.to(device), top_ks.to(device), top_ps.to(device), None,
                                             seeds.to(device), pos.to(device), return_n_keep=True)
    assert n_keep.cpu().tolist() == _kept_counts(kept).tolist()
    assert ids.cpu().tolist() == ref.tolist()


def test_unfiltered_seeded_sampling_matches_oracle(device):
    K = _k()
    g = torch.Generator().manual_seed(3)
    B, V = 8, 32000
    probs = torch.softmax(torch.randn((B, V), generator=g) * 2, dim=-1)
    seeds = torch.randint(0, 2 ** 62, (B,), generator=g)
    pos = torch.randint(0, 4096, (B,), generator=g)
    ref = oh.sampling_from_probs(probs, seeds, pos)
    ids = K.top_k_top_p_min_p_sample(probs.to(device), None, None, None, seeds.to(device), pos.to(device),
                                     filtered=False)
    assert ids.cpu().tolist() == ref.tolist()


def test_unfiltered_sampling_over_column_ranges_equals_the_row_kernel(device):
    """Decode-sized batches of wide rows: the unfiltered sampler cuts a row into column ranges over the whole chip (given
    its workspace).  Same token as the one-workgroup-per-row kernel and as the oracle, including a tie between two ranges
    (first index wins), a NaN (maximal) and an all-zero row."""
    K = _k()
    g = torch.Generator().manual_seed(11)
    B, V = 6, 128256
    probs = torch.softmax(torch.randn((B, V), generator=g) * 2, dim=-1)
    probs[1, 100] = probs[1, V - 5] = 0.9             # equal scores need equal gumbels too: rarely a tie -- the oracle decides
    probs[2, V // 2 + 3] = float("nan")
    probs[3] = 0.0
    seeds = torch.randint(0, 2 ** 62, (B,), generator=g)
    pos = torch.randint(0, 4096, (B,), generator=g)
    split = K.top_k_top_p_min_p_sample(probs.to(device), None, None, None, seeds.to(device), pos.to(device), filtered=False)
    whole = K.top_k_top_p_min_p_sample(probs.to(device), None, None, None, seeds.to(device), pos.to(device), filtered=False,
                                       use_workspace=False)
    assert split.cpu().tolist() == whole.cpu().tolist()
    ok = [0, 1, 4, 5]                                  # (rows 2 and 3: NaN / log(0) rows are compared kernel to kernel only)
    ref = oh.sampling_from_probs(probs[ok], seeds[ok], pos[ok])
    assert split.cpu()[ok].tolist() == ref.tolist()


@pytest.mark.parametrize("V,B", [(128256, 64), (32000, 7), (151936, 3), (16388, 20)])
def test_filtered_sampling_over_column_ranges_equals_the_row_kernel(device, V, B):
    """Round 6: the filtered sampler for decode-sized batches -- the first radix level's histogram and the collection of the cut's
    candidates as column ranges over the whole chip, a third launch finishing each row from its candidate list.  Ids AND kept counts
    must equal the one-workgroup-per-row kernel's (and through it the oracle's) at every range count: peaked rows (few candidates),
    flat rows (more candidates than the list holds: those rows run the whole routine in the finish launch), top-k only / top-p only /
    both / min-p, a tie that straddles a range boundary, top_k = 0."""
    K = _k()
    g = torch.Generator().manual_seed(V + B)
    scale = torch.tensor([4.0, 1.0, 8.0, 0.3, 2.0, 6.0, 3.0][:B] + [3.0] * max(0, B - 7)).unsqueeze(1)
    probs = torch.softmax(torch.randn((B, V), generator=g) * scale, dim=-1)
    if B >= 3:
        per = ((V + 7) // 8 + 3) // 4 * 4                  # a 4-way tie around the first boundary of the 8-range split, cut by top_k = 2
        probs[2] *= 0.2
        probs[2, [per - 2, per - 1, per, per + 1]] = 0.2
    top_ks = torch.tensor(([50, 1, 2, TOP_K_ALL, 1000, 0, 20] * (B // 7 + 1))[:B], dtype=torch.int32)
    top_ps = torch.tensor(([0.9, 1.0, 1.0, 0.95, 0.8, 0.5, 0.999] * (B // 7 + 1))[:B])
    min_ps = torch.tensor(([0.0, 0.0, 0.0, 0.01, 0.0, 0.0, 0.2] * (B // 7 + 1))[:B])
    seeds = torch.randint(0, 2 ** 62, (B,), generator=g)
    pos = torch.randint(0, 4096, (B,), generator=g)
    args = (probs.to(device), top_ks.to(device), top_ps.to(device))
    for mp in (None, min_ps.to(device)):
        want, want_n = K.top_k_top_p_min_p_sample(*args, mp, seeds.to(device), pos.to(device), return_n_keep=True, ranges=0)
        for ranges in (2, 5, 8, 16):
            got, got_n = K.top_k_top_p_min_p_sample(*args, mp, seeds.to(device), pos.to(device), return_n_keep=True, ranges=ranges)
            assert got_n.cpu().tolist() == want_n.cpu().tolist(), (ranges, mp is not None)
            assert got.cpu().tolist() == want.cpu().tolist(), (ranges, mp is not None)
    assert K.sample_ranges(64, 128256) == 8 and K.sample_ranges(512, 128256) == 0 and K.sample_ranges(64, 4096) == 0


@pytest.mark.parametrize("V,B", [(128256, 64), (128256, 16), (40000, 3), (65536, 130), (151936, 1)])
def test_sampling_straight_from_bf16_logits_equals_softmax_then_sample(device, V, B):
    """Round 6: Sampler.forward's filtered case for the bf16 logits of a decode-sized batch in ONE native call that never writes the
    probabilities: every column range selects its largest logits exactly, only those become probabilities (the softmax launches' own
    partials / merge / formula: the same bits) and are ranked, filtered and sampled; rows the candidates cannot decide are redone from
    their full probability row inside the call.  Ids AND kept counts must equal softmax_temperature_from_bf16 +
    top_k_top_p_min_p_sample on every row: peaked and flat rows, top-k only / top-p only / both / min-p, top_k in {0, 1, 64, 65, all},
    ties wider than a range's list, rows of near-zero logits under one large one (distinct logits, equal probabilities), hot and
    cold temperatures.  Rows with top_k <= 64 over gaussian logits must take the short way."""
    K = _k()
    g = torch.Generator().manual_seed(3 * V + B)
    scale = torch.tensor(([4.0, 1.0, 8.0, 0.3, 2.0, 6.0, 3.0, 0.05] * (B // 8 + 1))[:B]).unsqueeze(1)
    lg = torch.randn((B, V), generator=g) * scale
    temps = (torch.rand((B, 1), generator=g) + 0.5)
    top_ks = torch.tensor(([50, 1, 64, TOP_K_ALL, 65, 0, 20, 1000] * (B // 8 + 1))[:B], dtype=torch.int32)
    top_ps = torch.tensor(([0.9, 1.0, 1.0, 0.95, 0.8, 0.5, 0.999, 0.9] * (B // 8 + 1))[:B])
    min_ps = torch.tensor(([0.0, 0.0, 0.0, 0.01, 0.0, 0.0, 0.2, 0.0] * (B // 8 + 1))[:B])
    special = {}
    if B >= 16:
        lg[8] = 0.0                                            # every logit equal: a tie wider than any list (top_k 50)
        lg[9] = (torch.randn(V, generator=g) * 1e-6)           # near-zero logits under one large one: x / t - max collapses,
        lg[9, 77] = 12.0                                       # distinct logits with equal probabilities (top_k 1 / top_p 1)
        lg[10] = (torch.randint(0, 3, (V,), generator=g).float() - 1.0) * 4      # three values only: huge ties at the cut (top_k 64)
        lg[11, :] = -30.0
        lg[11, 5:9] = torch.tensor([2.0, 2.0, 1.0, 2.0])       # a tiny support (top_k all, top_p 0.95, min_p 0.01)
        temps[12] = 40.0                                       # hot: flat row, top_k 65 / top_p 0.8
        temps[14] = 0.05                                       # cold
        special = {8, 9, 10, 12}
    lg = lg.to(torch.bfloat16)
    seeds = torch.randint(0, 2 ** 62, (B,), generator=g)
    pos = torch.randint(0, 4096, (B,), generator=g)
    d = lambda x: x.to(device)
    if K.softmax_temperature_from_bf16(d(lg), d(temps)) is None:
        assert K.sample_from_logits(d(lg), d(temps), d(top_ks), d(top_ps), None, d(seeds), d(pos)) is None
        return
    for mp in (None, d(min_ps)):
        probs = K.softmax_temperature_from_bf16(d(lg), d(temps))
        want, want_n = K.top_k_top_p_min_p_sample(probs, d(top_ks), d(top_ps), mp, d(seeds), d(pos), return_n_keep=True)
        for src in (d(lg), d(lg).float()):               # bf16 logits; the same values as fp32 (the reference's LogitsProcessor widens them)
            before = src.clone()
            got, got_n, fb = K.sample_from_logits(src, d(temps), d(top_ks), d(top_ps), mp, d(seeds), d(pos), return_n_keep=True,
                                                  return_fallback=True)
            fb = fb.cpu().tolist()
            assert torch.equal(src, before)                 # the logits are read, never overwritten
            assert got_n.cpu().tolist() == want_n.cpu().tolist(), (mp is not None, str(src.dtype), fb)
            assert got.cpu().tolist() == want.cpu().tolist(), (mp is not None, str(src.dtype), fb)
            for r in range(B):
                if r not in special and 1 <= int(top_ks[r]) <= 64 and float(scale[r]) >= 1.0:
                    assert fb[r] == 0, (r, fb)
            assert all(fb[r] == 1 for r in range(B) if int(top_ks[r]) == 0)
    # genuine fp32 logits (low mantissa bits in use): against the fp32 softmax + sample
    lf = (d(lg).float() + d(torch.randn((B, V), generator=g)) * 1e-3).contiguous()
    probs = K.softmax_temperature_(lf.clone(), d(temps))
    want, want_n = K.top_k_top_p_min_p_sample(probs, d(top_ks), d(top_ps), None, d(seeds), d(pos), return_n_keep=True)
    got, got_n = K.sample_from_logits(lf, d(temps), d(top_ks), d(top_ps), None, d(seeds), d(pos), return_n_keep=True)
    assert got_n.cpu().tolist() == want_n.cpu().tolist() and got.cpu().tolist() == want.cpu().tolist()
    # the split rule of the softmax it shares its partials with; other range counts have no short way
    assert K.sample_from_bf16_logits(d(lg[:1, :4096].contiguous()), d(temps[:1]), d(top_ks[:1]), d(top_ps[:1]), None, d(seeds[:1]),
                                     d(pos[:1])) is None or V < 4096 * 8


def test_sampling_from_bf16_logits_unsupported_range_counts_fall_back(device):
    """B = 40 -> 51 softmax ranges (neither <= 16 nor a multiple of 16): no short way, the Sampler module takes the two calls."""
    K = _k()
    lg = torch.randn((40, 128256)).to(torch.bfloat16).to(device)
    t = torch.ones((40, 1), device=device)
    assert K.sample_from_bf16_logits(lg, t, None, torch.full((40,), 0.9, device=device), None, torch.arange(40, device=device), None) is None
    assert K.softmax_temperature_from_bf16(lg, t) is not None


def test_ties_and_degenerate_rows(device):
    K = _k()
    V = 4096
    probs = torch.zeros((4, V))
    probs[0, :8] = 1 / 8                      # 8-way tie, top_k = 3 -> exactly 3 kept
    probs[1, 5] = 1.0                         # one-hot
    probs[2] = 1.0 / V                        # uniform, top_p 0.25 -> V/4 kept (+1: exclusive cumsum)
    probs[3, 100:110] = torch.tensor([0.3, 0.2, 0.2, 0.1, 0.05, 0.05, 0.04, 0.03, 0.02, 0.01])
    top_ks = torch.tensor([3, TOP_K_ALL, TOP_K_ALL, 0], dtype=torch.int32)
    top_ps = torch.tensor([1.0, 0.5, 0.25, 1.0])
    seeds = torch.tensor([1, 2, 3, 4])
    pos = torch.tensor([9, 8, 7, 6])
    ids, n_keep = K.top_k_top_p_min_p_sample(probs.to(device), top_ks.to(device), top_ps.to(device), None,
                                             seeds.to(device), pos.to(device), return_n_keep=True)
    n = n_keep.cpu().tolist()
    assert n[0] == 3 and n[1] == 1 and abs(n[2] - (V // 4 + 1)) <= 1 and n[3] == 0
    ids = ids.cpu().tolist()
    assert ids[0] in range(8) and ids[1] == 5 and 0 <= ids[2] < V
    assert ids[3] == 100                      # empty nucleus: the reference argmax-es to sorted rank 0


def test_sampling_distribution(device):
    """Unseeded sampling draws from the renormalised kept distribution (chi-square style bound)."""
    K = _k()
    V, N = 512, 20000
    g = torch.Generator().manual_seed(0)
    p = torch.softmax(torch.randn(V, generator=g) * 2, dim=-1)
    probs = p.repeat(N, 1).to(device)
    top_k = 8
    ids = K.top_k_top_p_min_p_sample(probs, torch.full((N,), top_k, dtype=torch.int32, device=device),
                                     torch.ones(N, device=device), None, None, None)
    top = torch.topk(p, top_k)
    want = top.values / top.values.sum()
    counts = torch.bincount(ids.cpu().long(), minlength=V).float()
    assert counts.sum() == N and counts[top.indices].sum() == N
    freq = counts[top.indices] / N
    assert (freq - want).abs().max() < 0.02


def test_renorm_probs_match_torch(device):
    """sampler.py:753-762 top_p_normalize_probs_torch and the top-k analogue."""
    K = _k()
    g = torch.Generator().manual_seed(7)
    B, V = 8, 32000
    probs = torch.softmax(torch.randn((B, V), generator=g) * 3, dim=-1)
    top_ps = torch.tensor([0.9, 0.5, 0.99, 0.1, 0.7, 0.3, 0.8, 0.95])
    srt, idx = probs.sort(dim=-1, descending=True)
    cs = torch.cumsum(srt, dim=-1)
    srt_p = srt.clone()
    srt_p[(cs - srt) > top_ps.view(-1, 1)] = 0.0
    srt_p.div_(srt_p.sum(dim=-1, keepdim=True))
    want_p = torch.zeros_like(srt_p).scatter_(-1, idx, srt_p)
    got_p = K.top_p_renorm_prob(probs.to(device), top_ps.to(device)).cpu()
    assert torch.equal(got_p > 0, want_p > 0)
    torch.testing.assert_close(got_p, want_p, rtol=1e-5, atol=1e-9)
    top_ks = torch.tensor([1, 5, 50, 1000, 7, 64, 2, 300], dtype=torch.int32)
    srt_k = srt.clone()
    srt_k[torch.arange(V).view(1, -1) >= top_ks.view(-1, 1)] = 0.0
    srt_k.div_(srt_k.sum(dim=-1, keepdim=True))
    want_k = torch.zeros_like(srt_k).scatter_(-1, idx, srt_k)
    got_k = K.top_k_renorm_prob(probs.to(device), top_ks.to(device)).cpu()
    assert torch.equal(got_k > 0, want_k > 0)
    torch.testing.assert_close(got_k, want_k, rtol=1e-5, atol=1e-9)
    got_s = K.top_k_renorm_prob(probs.to(device), 10).cpu()
    assert ((got_s > 0).sum(dim=1) == 10).all()


def test_sampler_module_flow(device):
    """Sampler.forward: greedy, simple seeded sampling and filtered seeded sampling vs the oracle."""
    from sglang_amd.layers.sampler import LogitsProcessorOutput, Sampler, SamplingBatchInfo

    g = torch.Generator().manual_seed(21)
    B, V = 8, 32000
    logits = torch.randn((B, V), generator=g) * 3
    temps = torch.rand((B, 1), generator=g) + 0.5
    top_ks = torch.tensor([20, 50, 1, 5, 40, 10, 30, 64], dtype=torch.int32)
    top_ps = torch.tensor([0.9, 0.8, 1.0, 0.5, 0.95, 0.6, 0.7, 0.85])
    seeds = torch.randint(0, 2 ** 62, (B,), generator=g)
    pos = torch.randint(0, 4096, (B,), generator=g)
    smp = Sampler()
    greedy = smp(LogitsProcessorOutput(logits.to(device).clone()), SamplingBatchInfo.greedy(B, device))
    assert greedy.cpu().tolist() == logits.argmax(-1).tolist()
    probs = torch.softmax(logits / temps, dim=-1)
    info = SamplingBatchInfo(temps.to(device), top_ps.to(device), top_ks.to(device), torch.zeros(B, device=device),
                             is_all_greedy=False, need_top_p_sampling=True, need_top_k_sampling=True,
                             sampling_seed=seeds.to(device))
    got = smp(LogitsProcessorOutput(logits.to(device).clone()), info, positions=pos.to(device))
    # the GPU softmax differs from torch's CPU softmax in the last ulp; compare on the GPU's own probs
    from sglang_amd import kernels as K

    gp = K.softmax_temperature_(logits.to(device).clone(), temps.to(device)).cpu()
    torch.testing.assert_close(gp, probs, rtol=2e-5, atol=1e-9)
    ref = oh.top_k_top_p_min_p_sampling_from_probs(gp.clone(), top_ks, top_ps, None, False, seeds, pos)
    assert got.cpu().tolist() == ref.tolist()
    info2 = SamplingBatchInfo(temps.to(device), torch.ones(B, device=device),
                              torch.full((B,), TOP_K_ALL, dtype=torch.int32, device=device),
                              torch.zeros(B, device=device), is_all_greedy=False, sampling_seed=seeds.to(device))
    got2 = smp(LogitsProcessorOutput(logits.to(device).clone()), info2, positions=pos.to(device))
    assert got2.cpu().tolist() == oh.sampling_from_probs(gp, seeds, pos).tolist()


def test_softmax_from_bf16_logits_equals_widen_then_softmax(device):
    """Round 6: the model-dtype logits of a decode-sized batch are widened INSIDE the two softmax launches: bit-identical to
    `logits.float()` followed by the fp32 kernel (the widening is exact, the arithmetic and the range order the same), and the
    Sampler module takes that route for bf16 logits (same ids as with pre-widened logits)."""
    from sglang_amd.layers.sampler import LogitsProcessorOutput, Sampler, SamplingBatchInfo

    K = _k()
    g = torch.Generator().manual_seed(5)
    assert K.softmax_temperature_from_bf16(torch.zeros((3, 32000), dtype=torch.bfloat16, device=device), torch.ones((3, 1), device=device)) is None
    for B, V in ((64, 128256), (3, 40000), (17, 151936)):
        lg = (torch.randn((B, V), generator=g) * 3).to(torch.bfloat16).to(device)
        temps = (torch.rand((B, 1), generator=g) + 0.5).to(device)
        want = K.softmax_temperature_(lg.float(), temps)
        got = K.softmax_temperature_from_bf16(lg, temps)
        assert got is not None and torch.equal(got, want), (B, V)
    B, V = 16, 128256
    lg = (torch.randn((B, V), generator=g) * 2).to(torch.bfloat16).to(device)
    info = SamplingBatchInfo(torch.ones((B, 1), device=device), torch.full((B,), 0.9, device=device),
                             torch.full((B,), 50, dtype=torch.int32, device=device), torch.zeros(B, device=device), False,
                             need_top_p_sampling=True, need_top_k_sampling=True,
                             sampling_seed=torch.arange(B, device=device, dtype=torch.int64) + 7)
    pos = torch.arange(B, device=device, dtype=torch.int64)
    smp = Sampler()
    a = smp(LogitsProcessorOutput(next_token_logits=lg), info, positions=pos)
    b = smp(LogitsProcessorOutput(next_token_logits=lg.float()), info, positions=pos)
    assert a.cpu().tolist() == b.cpu().tolist()
