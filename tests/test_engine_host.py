"""The engine's host logic on the CPU: radix hits, slot / row bookkeeping, the overlap-scheduled token hand-off and
the release of everything at the end, driven by a stand-in model runner (a deterministic toy LM) with CPU pools.
The four integer helper kernels the engine calls are replaced by their oracle restatements (oracle/host.py); the
GPU suite (tests/test_engine_gpu.py) runs the same class with the real runner."""
import ctypes
import random

import pytest
import torch

from oracle import host as oh
from sglang_amd import kernels
from sglang_amd.harness.engine import Engine, Req
from sglang_amd.layers.sampler import LogitsProcessorOutput
from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator
from sglang_amd.mem_cache.memory_pool import ReqToTokenPool
from sglang_amd.mem_cache.radix_cache import RadixCache

VOCAB = 97


def _toy_next(last_token: int, kv_len: int) -> int:
    return (last_token * 7 + kv_len * 3 + 1) % VOCAB


class _Pool:
    def __init__(self, size):
        self.size = size


class _ToyRunner:
    """What Engine reads of harness.engine.ModelRunner, with a next-token rule that depends on the last input
    token and the number of KV rows the request has (so a wrong seq_len or a stale last token shows up)."""

    def __init__(self, max_reqs, ctx, size, disable_radix=False, page_size=1):
        self.device = torch.device("cpu")
        self.page_size = page_size
        self.req_to_token_pool = ReqToTokenPool(max_reqs, ctx, self.device)
        self.token_to_kv_pool = _Pool(size)
        self.token_to_kv_pool_allocator = (
            TokenToKVPoolAllocator(size, torch.bfloat16, self.device, None) if page_size == 1
            else PagedTokenToKVPoolAllocator(size, page_size, torch.bfloat16, self.device, None))
        self.tree_cache = RadixCache(self.req_to_token_pool, self.token_to_kv_pool_allocator, page_size,
                                     disable=disable_radix)
        self.attn_backend = None
        self.graph_runner = None
        self.seen = []                                   # (mode, kv lengths, slots written) per forward

    def forward(self, fb):
        if fb.forward_mode.is_extend():
            last = torch.cumsum(torch.tensor(fb.extend_seq_lens_cpu), 0) - 1
            last_tok = fb.input_ids[last].tolist()
        else:
            last_tok = fb.input_ids.tolist()
        kv = fb.seq_lens.tolist()
        self.seen.append((fb.forward_mode.is_extend(), kv, fb.out_cache_loc.tolist(), fb.positions.tolist()))
        logits = torch.full((len(kv), VOCAB), -1.0)
        for b, (t, n) in enumerate(zip(last_tok, kv)):
            logits[b, _toy_next(t, n)] = 1.0
        return LogitsProcessorOutput(next_token_logits=logits)

    def sample(self, logits_output, fb):
        return logits_output.next_token_logits.argmax(-1)


@pytest.fixture
def cpu_kernels(monkeypatch):
    def write_req_to_token(r2t, req_pool, prefix_ptrs, prefix_lens, seq_lens, ext_lens, out_cache_loc):
        off = 0
        for b in range(req_pool.numel()):
            row, pre, seq, ext = int(req_pool[b]), int(prefix_lens[b]), int(seq_lens[b]), int(ext_lens[b])
            if pre and int(prefix_ptrs[b]):          # a null pointer keeps the row's prefix (mixed batches)
                src = (ctypes.c_int64 * pre).from_address(int(prefix_ptrs[b]))
                r2t[row, :pre] = torch.tensor(list(src), dtype=torch.int32)
            r2t[row, pre:seq] = out_cache_loc[off: off + ext].to(torch.int32)
            off += ext

    def compute_position(prefix_lens, extend_lens, total):
        pos, start = oh.compute_position(prefix_lens.tolist(), extend_lens.tolist())
        return torch.from_numpy(pos).to(torch.int64), torch.from_numpy(start).to(extend_lens.dtype)

    def clamp_position(seq_lens, out=None):
        res = torch.from_numpy(oh.clamp_position(seq_lens.tolist())).to(torch.int64)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def alloc_extend(prefix_lens, seq_lens, last_loc, free_pages, out_indices, page_size):
        idx, _ = oh.alloc_extend(prefix_lens.tolist(), seq_lens.tolist(), last_loc.tolist(), free_pages.tolist(), page_size)
        out_indices.copy_(torch.from_numpy(idx))

    def alloc_decode(seq_lens, last_loc, free_pages, out_indices, page_size):
        idx, _ = oh.alloc_decode(seq_lens.tolist(), last_loc.tolist(), free_pages.tolist(), page_size)
        out_indices.copy_(torch.from_numpy(idx))

    monkeypatch.setattr(kernels, "alloc_extend", alloc_extend)
    monkeypatch.setattr(kernels, "alloc_decode", alloc_decode)
    monkeypatch.setattr(kernels, "write_req_to_token", write_req_to_token)
    monkeypatch.setattr(kernels, "compute_position", compute_position)
    monkeypatch.setattr(kernels, "clamp_position", clamp_position)


def _prompts(groups, per_group, shared, seed=3):
    rnd = random.Random(seed)
    out = []
    for _ in range(groups):
        sys_p = [rnd.randrange(VOCAB) for _ in range(shared)]
        out += [sys_p + [rnd.randrange(VOCAB) for _ in range(rnd.randrange(2, 9))] for _ in range(per_group)]
    return out


def _expected(prompt, new_tokens):
    toks, out = list(prompt), []
    for _ in range(new_tokens):
        out.append(_toy_next(toks[-1], len(toks)))
        toks.append(out[-1])
    return out


@pytest.mark.parametrize("disable_radix", [False, True])
@pytest.mark.parametrize("lag", [0, 1])
def test_engine_bookkeeping_with_a_toy_model(cpu_kernels, disable_radix, lag):
    groups, per_group, shared, new_tokens = 3, 4, 20, 6
    prompts = _prompts(groups, per_group, shared)
    B = len(prompts)
    size = B * 64
    runner = _ToyRunner(B, 64, size, disable_radix)
    eng = Engine(runner)
    reqs = [Req(i, p, new_tokens) for i, p in enumerate(prompts)]
    leaders = [q for q in reqs if q.rid % per_group == 0]
    rest = [q for q in reqs if q.rid % per_group]
    eng.prefill(leaders)
    eng.prefill(rest)
    # radix hits: followers reuse their leader's shared prefix (the whole system prompt)
    assert [q.cached_tokens for q in leaders] == [0] * groups
    assert [q.cached_tokens for q in rest] == [0 if disable_radix else shared] * len(rest)
    r2t = runner.req_to_token_pool.req_to_token
    if not disable_radix:
        for q in rest:
            lead = reqs[q.rid - q.rid % per_group]
            assert torch.equal(r2t[q.req_pool_idx, :shared], r2t[lead.req_pool_idx, :shared])
    rows = [r2t[q.req_pool_idx, : len(q.origin_input_ids)].tolist() for q in reqs]
    private = [s for q, row in zip(reqs, rows) for s in (row if q.rid % per_group == 0 or disable_radix else row[shared:])]
    assert len(set(private)) == len(private) and 0 not in private      # every computed token owns a distinct slot
    # extend positions start at the cached prefix
    ext_calls = [c for c in runner.seen if c[0]]
    assert ext_calls[1][3][:3] == [0 if disable_radix else shared, (0 if disable_radix else shared) + 1,
                                   (0 if disable_radix else shared) + 2]
    for _ in range(new_tokens - 1):
        eng.decode_step()
        eng.flush_decode_outputs(lag=lag)
    done = sorted(eng.running, key=lambda q: q.rid)
    eng.finish(list(eng.running))
    for q in done:
        assert q.output_ids == _expected(q.origin_input_ids, new_tokens), q.rid
    # decode wrote one new slot per request per step, at column kv_len - 1 of its row, positions = kv_len - 1
    dec_calls = [c for c in runner.seen if not c[0]]
    assert len(dec_calls) == new_tokens - 1
    for step, (_, kv, slots, pos) in enumerate(dec_calls):
        assert len(set(slots)) == B and pos == [n - 1 for n in kv]
        assert kv == [len(q.origin_input_ids) + step + 1 for q in eng_order(leaders, rest)]
    # nothing leaked
    tree, alloc = runner.tree_cache, runner.token_to_kv_pool_allocator
    assert tree.protected_size() == 0
    assert alloc.available_size() + tree.evictable_size() == size
    assert runner.req_to_token_pool.available_size() == runner.req_to_token_pool.size


def eng_order(leaders, rest):
    return list(leaders) + list(rest)


def test_results_do_not_depend_on_the_radix_cache(cpu_kernels):
    prompts = _prompts(2, 5, 16, seed=8)
    outs = []
    for disable in (False, True):
        runner = _ToyRunner(len(prompts), 64, len(prompts) * 64, disable)
        eng = Engine(runner)
        reqs = [Req(i, p, 5) for i, p in enumerate(prompts)]
        eng.generate(reqs, sync_every=2)
        outs.append([q.output_ids for q in reqs])
    assert outs[0] == outs[1] == [_expected(p, 5) for p in prompts]


@pytest.mark.parametrize("page", [4, 16])
def test_paged_engine_bookkeeping(cpu_kernels, page):
    """page_size > 1: radix hits are page aligned, a request's tokens fill whole pages in order
    (allocator/paged.py:45-102 through the oracle's restatement), decode opens a page every `page` tokens."""
    groups, per_group, shared, new_tokens = 2, 3, 37, 9
    prompts = _prompts(groups, per_group, shared, seed=5)
    B = len(prompts)
    size = B * 96 // page * page
    runner = _ToyRunner(B, 96, size, page_size=page)
    eng = Engine(runner)
    reqs = [Req(i, p, new_tokens) for i, p in enumerate(prompts)]
    leaders = [q for q in reqs if q.rid % per_group == 0]
    rest = [q for q in reqs if q.rid % per_group]
    eng.prefill(leaders)
    eng.prefill(rest)
    assert [q.cached_tokens for q in rest] == [shared // page * page] * len(rest)
    r2t = runner.req_to_token_pool.req_to_token
    for q in reqs:
        row = r2t[q.req_pool_idx, : len(q.origin_input_ids)].tolist()
        assert all(b == a + 1 for a, b, i in zip(row, row[1:], range(1, len(row))) if i % page)   # pages are contiguous
    for _ in range(new_tokens - 1):
        eng.decode_step()
        eng.flush_decode_outputs(lag=1)
    done = sorted(eng.running, key=lambda q: q.rid)
    for q in done:                                                    # rows stay page-contiguous while decoding
        row = r2t[q.req_pool_idx, : q.seqlen - 1].tolist()
        assert all(b == a + 1 for a, b, i in zip(row, row[1:], range(1, len(row))) if i % page)
    eng.finish(list(eng.running))
    for q in done:
        assert q.output_ids == _expected(q.origin_input_ids, new_tokens)
    tree, alloc = runner.tree_cache, runner.token_to_kv_pool_allocator
    assert tree.protected_size() == 0
    assert alloc.available_size() + tree.evictable_size() == size


@pytest.mark.parametrize("page", [1, 4])
@pytest.mark.parametrize("chunk", [16, 24, 64, 1000])
def test_chunked_prefill_matches_whole_prompt_prefill(cpu_kernels, page, chunk):
    """prefill_chunked (schedule_policy.py:1004-1060 / 1160-1200): every pass stays inside the token budget, at most
    one request is truncated per pass and it continues first, and the generated tokens / final cache state are those
    of the unchunked prefill."""
    chunk = max(chunk // page * page, page)
    prompts = _prompts(2, 3, 30, seed=11) + [[(3 * i + 1) % VOCAB for i in range(70)]]      # one long prompt
    B, new_tokens = len(prompts), 4
    size = B * 128 // page * page
    runner = _ToyRunner(B, 128, size, page_size=page)
    eng = Engine(runner)
    reqs = [Req(i, p, new_tokens) for i, p in enumerate(prompts)]
    eng.prefill_chunked(reqs, chunk)
    ext = [c for c in runner.seen if c[0]]
    assert all(len(slots) <= chunk for _, _, slots, _ in ext)                   # the budget holds in every pass
    assert sum(len(slots) for _, _, slots, _ in ext) == sum(len(p) - q.cached_tokens for p, q in zip(prompts, reqs))
    if chunk < 70:
        assert len(ext) >= 2                                                    # the long prompt needed several passes
    assert sorted(q.rid for q in eng.running) == list(range(B))
    for _ in range(new_tokens - 1):
        eng.decode_step()
        eng.flush_decode_outputs(lag=1)
    done = sorted(eng.running, key=lambda q: q.rid)
    eng.finish(list(eng.running))
    for q in done:
        assert q.output_ids == _expected(q.origin_input_ids, new_tokens), q.rid
    tree, alloc = runner.tree_cache, runner.token_to_kv_pool_allocator
    assert tree.protected_size() == 0
    assert alloc.available_size() + tree.evictable_size() == size
    assert runner.req_to_token_pool.available_size() == runner.req_to_token_pool.size


def test_waves_of_requests_reuse_the_tree_and_the_slots(cpu_kernels):
    """Two generations back to back on one engine: the second wave hits what the first one left in the radix
    tree (prompt + generated tokens of a finished request are cached), under a pool small enough to need eviction."""
    prompts = _prompts(2, 3, 24, seed=21)
    B = len(prompts)
    size = sum(len(p) + 14 for p in prompts) * 4 // 3         # holds one wave, not both: the second one must evict
    runner = _ToyRunner(B, 96, size)
    eng = Engine(runner)
    first = [Req(i, p, 5) for i, p in enumerate(prompts)]
    eng.generate(first)
    assert [q.output_ids for q in first] == [_expected(p, 5) for p in prompts]
    # second wave: each prompt continued by its own generated tokens -> the whole old sequence but the last token
    # (whose KV row was never written) is a radix hit
    second_prompts = [p + q.output_ids + [5, 6, 7] for p, q in zip(prompts, first)]
    second = [Req(100 + i, p, 4) for i, p in enumerate(second_prompts)]
    eng.generate(second)
    assert [q.output_ids for q in second] == [_expected(p, 4) for p in second_prompts]
    hits = [q.cached_tokens for q in second]
    assert all(h >= len(p) + 5 - 1 - 24 for h, p in zip(hits, prompts)) and max(hits) >= 24
    tree, alloc = runner.tree_cache, runner.token_to_kv_pool_allocator
    assert tree.protected_size() == 0 and alloc.available_size() + tree.evictable_size() == size


@pytest.mark.parametrize("lag", [0, 1])
def test_prefill_between_lagged_decode_steps_keeps_every_token(cpu_kernels, lag):
    """A new request joins the running batch while hand-offs of the previous steps are still in flight
    (flush_decode_outputs(lag=1) then prefill): the in-flight tokens must reach their requests before the
    per-batch decode state is rebuilt, every request ends with exactly max_new_tokens correct tokens, and no
    KV slot leaks (a stale seq_len would overwrite a req_to_token column)."""
    prompts = _prompts(2, 3, 12, seed=21)
    first, late = prompts[:4], prompts[4:]
    n_first, n_late = 9, 5
    runner = _ToyRunner(len(prompts), 64, len(prompts) * 64)
    eng = Engine(runner)
    a = [Req(i, p, n_first) for i, p in enumerate(first)]
    b = [Req(10 + i, p, n_late) for i, p in enumerate(late)]
    eng.prefill(a)
    for _ in range(3):
        eng.decode_step()
        eng.flush_decode_outputs(lag=lag)
    eng.prefill(b)                                  # joins between two lagged steps
    for _ in range(n_late - 1):
        eng.decode_step()
        eng.flush_decode_outputs(lag=lag)
    eng.finish(b)                                   # the late ones are done; the first ones go on
    for _ in range(n_first - 1 - 3 - (n_late - 1)):
        eng.decode_step()
        eng.flush_decode_outputs(lag=lag)
    eng.finish(list(eng.running))
    for q in a + b:
        assert q.output_ids == _expected(q.origin_input_ids, q.max_new_tokens), q.rid
    tree, alloc = runner.tree_cache, runner.token_to_kv_pool_allocator
    assert tree.protected_size() == 0
    assert alloc.available_size() + tree.evictable_size() == len(prompts) * 64
    assert runner.req_to_token_pool.available_size() == runner.req_to_token_pool.size


def test_mixed_step_prefills_new_requests_and_advances_the_running_ones(cpu_kernels):
    """ForwardMode.MIXED: the running requests join the extend batch as 1-token extends over their cached rows."""
    prompts = _prompts(2, 3, 10, seed=31)
    runner = _ToyRunner(len(prompts), 64, len(prompts) * 64)
    eng = Engine(runner)
    a = [Req(i, prompts[i], 7) for i in (0, 1, 3)]           # two of group 0, one of group 1
    b = [Req(10 + i, prompts[i], 4) for i in (2, 4, 5)]      # the others: every one finds its group's prefix cached
    eng.prefill(a)
    eng.decode_step(); eng.flush_decode_outputs(lag=1)
    eng.decode_step()                                       # one hand-off still in flight when the mixed batch forms
    eng.mixed_step(b)
    mixed = runner.seen[-1]
    assert mixed[0] and len(mixed[1]) == 6                  # an extend-mode forward over 3 new + 3 running requests
    assert mixed[1][3:] == [len(q.origin_input_ids) + 3 for q in a]          # kv lengths of the running ones after the step
    assert [q.cached_tokens for q in b] == [10, 10, 10]     # the group's prefix was cached by the first batch
    for _ in range(3):
        eng.decode_step(); eng.flush_decode_outputs(lag=1)
    eng.finish(list(eng.running))
    for q in a:
        assert q.output_ids == _expected(q.origin_input_ids, 7), q.rid
    for q in b:
        assert q.output_ids == _expected(q.origin_input_ids, 4), q.rid
    tree, alloc = runner.tree_cache, runner.token_to_kv_pool_allocator
    assert tree.protected_size() == 0 and alloc.available_size() + tree.evictable_size() == len(prompts) * 64


def test_retraction_under_pool_pressure_keeps_every_token(cpu_kernels):
    """A KV pool too small for the whole batch's decode: the requests with the fewest generated tokens (longest
    prompts first among equals) are retracted, their tokens folded into their prompts, re-prefilled later, and every
    request still ends with exactly its expected tokens."""
    rnd = random.Random(5)
    prompts = [[rnd.randrange(VOCAB) for _ in range(n)] for n in (20, 26, 23, 29)]
    new_tokens = 12
    size = 20 + 26 + 23 + 29 + 4 * 3              # room for three decode steps of all four, then pressure
    runner = _ToyRunner(4, 64, size, disable_radix=True)
    eng = Engine(runner)
    reqs = [Req(i, p, new_tokens) for i, p in enumerate(prompts)]
    orig = [list(p) for p in prompts]
    eng.prefill(reqs)
    steps = 0
    while eng.running or eng.waiting:
        if not eng.running or (eng.waiting and eng._fits_with(eng.waiting[0])):
            q = eng.waiting.pop(0)
            eng.prefill([q])
        done = [q for q in eng.running if q.finished()]
        if done:
            eng.finish(done)
            continue
        eng.decode_step()
        eng.flush_decode_outputs()
        steps += 1
        assert steps < 200
    assert eng.stats.get("retracted", 0) >= 1
    for q, p in zip(reqs, orig):
        assert q.all_output_ids == _expected(p, new_tokens), q.rid
    alloc, tree = runner.token_to_kv_pool_allocator, runner.tree_cache
    assert alloc.available_size() + tree.evictable_size() == size and runner.req_to_token_pool.available_size() == 4


def test_generate_drains_retracted_requests_and_keeps_sampling_rows_with_their_requests(cpu_kernels):
    """Engine.generate() under pool pressure: requests retracted by a decode step are prefilled again and finished (none
    is dropped), and the per-batch sampling parameters follow the batch through every change of its composition -- each
    forward sees exactly the rows of the requests it runs, in their order."""
    from sglang_amd.layers.sampler import SamplingBatchInfo

    rnd = random.Random(6)
    prompts = [[rnd.randrange(VOCAB) for _ in range(n)] for n in (20, 26, 23, 29)]
    new_tokens = 12
    size = 20 + 26 + 23 + 29 + 4 * 3
    runner = _ToyRunner(4, 64, size, disable_radix=True)
    seen_rows = []

    def sample(logits_output, fb):
        info = fb.sampling_info
        assert info is not None and info.top_ps.numel() == logits_output.next_token_logits.shape[0]
        seen_rows.append((fb.batch_size if hasattr(fb, "batch_size") else None, info.top_ks.tolist()))
        return logits_output.next_token_logits.argmax(-1)

    runner.sample = sample
    eng = Engine(runner)
    reqs = [Req(i, p, new_tokens) for i, p in enumerate(prompts)]
    orig = [list(p) for p in prompts]
    info = SamplingBatchInfo.greedy(4, torch.device("cpu"))
    info.top_ks = torch.tensor([11, 22, 33, 44], dtype=torch.int32)        # a tag per request
    eng.generate(reqs, info)
    assert eng.stats.get("retracted", 0) >= 1 and not eng.running and not eng.waiting
    for q, p in zip(reqs, orig):
        assert q.all_output_ids == _expected(p, new_tokens), q.rid
    # batches of fewer than four requests ran, each with its own requests' tags
    assert any(len(tags) < 4 for _, tags in seen_rows)
    assert all(set(tags) <= {11, 22, 33, 44} and len(set(tags)) == len(tags) for _, tags in seen_rows)
    alloc, tree = runner.token_to_kv_pool_allocator, runner.tree_cache
    assert alloc.available_size() + tree.evictable_size() == size and runner.req_to_token_pool.available_size() == 4


def test_decode_memory_check_counts_the_steps_still_in_flight(cpu_kernels):
    """With hand-offs in flight (lag 1) output_ids trail the device by a step: the page-boundary test of
    new_tokens_required_next_decode must use the decode state's lengths, not len(output_ids)."""
    prompts = _prompts(1, 3, 7, seed=2)
    runner = _ToyRunner(3, 64, 3 * 64, page_size=4)
    eng = Engine(runner)
    reqs = [Req(i, p, 9) for i, p in enumerate(prompts)]
    eng.prefill(reqs)
    for _ in range(5):
        eng.decode_step()
        eng.flush_decode_outputs(lag=1)
        true_lens = [int(x) for x in eng._decode_state["seq_lens_cpu"].tolist()]
        assert eng._kv_lens() == true_lens
        assert eng.new_tokens_required_next_decode() == sum(1 for n in true_lens if n % 4 == 0) * 4
        assert any(q.seqlen - 1 != n for q, n in zip(eng.running, true_lens))      # the stale figure differs
