"""Host data model: the product RadixCache / allocators against (a) the trace
recorded from the real reference RadixCache (tests/golden/radix_trace.json),
(b) the reference's own unit-test scenarios, (c) the brute-force oracle.  CPU."""
import json
import random
from array import array
from unittest import mock

import pytest
import torch

from oracle.host import BruteForcePrefixCache
from sglang_amd.mem_cache.allocator import TokenToKVPoolAllocator, PagedTokenToKVPoolAllocator
from sglang_amd.mem_cache.radix_cache import (EvictParams, InsertParams, MatchPrefixParams, RadixCache, RadixKey)


def _key(ids):
    return RadixKey(array("q", ids))


@pytest.mark.parametrize("page", [1, 4])
def test_replay_reference_trace(golden_dir, page):
    ops = json.loads((golden_dir / "radix_trace.json").read_text())[f"page{page}"]
    alloc = mock.Mock()
    alloc.device = "cpu"
    tree = RadixCache.create_simulated(mock_allocator=alloc, page_size=page)
    held = []
    for i, op in enumerate(ops):
        if op["op"] == "insert":
            res = tree.insert(InsertParams(key=_key(op["ids"]), value=torch.tensor(op["vals"], dtype=torch.int64)))
            assert res.prefix_len == op["prefix_len"], (i, op)
        elif op["op"] == "match":
            m = tree.match_prefix(MatchPrefixParams(key=_key(op["ids"])))
            assert m.device_indices.tolist() == op["indices"], (i, op)
            assert m.device_indices.dtype == torch.int64
            if op["lock"]:
                tree.inc_lock_ref(m.last_device_node)
                held.append(m.last_device_node)
        elif op["op"] == "unlock":
            tree.dec_lock_ref(held.pop(op["which"]))
        else:
            alloc.reset_mock()
            res = tree.evict(EvictParams(num_tokens=op["num_tokens"]))
            assert res.num_tokens_evicted == op["num_evicted"], (i, op)
            freed = [c.args[0].tolist() for c in alloc.free_segment.call_args_list]
            assert freed == op["freed"], (i, op)
        assert tree.evictable_size() == op["evictable"], (i, op)
        assert tree.protected_size() == op["protected"], (i, op)
        assert tree.total_size() == op["total"], (i, op)


@pytest.mark.parametrize("seed,steps,vocab", [(7, 600, 6), (31, 600, 3), (1234, 900, 12)])
def test_live_differential_traces(golden_dir, tmp_path, seed, steps, vocab):
    """Not one recorded trace: where the reference's sources are present (the build container), a fresh process drives the REAL
    `RadixCache` through a new random sequence of insert / match (+ lock) / unlock / evict operations (tests/golden/gen_golden.py
    gen_radix with another seed, length, vocabulary; page sizes 1 / 4 / 16) and the product's tree must reproduce every returned
    prefix length, slot list, freed segment and size counter."""
    import subprocess
    import sys

    gen = golden_dir / "gen_golden.py"
    if not (__import__("pathlib").Path("/root/reference/python/sglang").exists()):
        pytest.skip("/root/reference not present: the live trace needs the reference's own RadixCache")
    out = tmp_path / "trace.json"
    p = subprocess.run([sys.executable, str(gen), "--radix-live", str(seed), str(steps), str(vocab), str(out)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    traces = json.loads(out.read_text())
    for page in (1, 4, 16):
        ops = traces[f"page{page}"]
        assert len(ops) == steps
        alloc = mock.Mock()
        alloc.device = "cpu"
        tree = RadixCache.create_simulated(mock_allocator=alloc, page_size=page)
        held = []
        for i, op in enumerate(ops):
            if op["op"] == "insert":
                res = tree.insert(InsertParams(key=_key(op["ids"]), value=torch.tensor(op["vals"], dtype=torch.int64)))
                assert res.prefix_len == op["prefix_len"], (page, i, op)
            elif op["op"] == "match":
                m = tree.match_prefix(MatchPrefixParams(key=_key(op["ids"])))
                assert m.device_indices.tolist() == op["indices"], (page, i, op)
                if op["lock"]:
                    tree.inc_lock_ref(m.last_device_node)
                    held.append(m.last_device_node)
            elif op["op"] == "unlock":
                tree.dec_lock_ref(held.pop(op["which"]))
            else:
                alloc.reset_mock()
                res = tree.evict(EvictParams(num_tokens=op["num_tokens"]))
                assert res.num_tokens_evicted == op["num_evicted"], (page, i, op)
                assert [c.args[0].tolist() for c in alloc.free_segment.call_args_list] == op["freed"], (page, i, op)
            assert (tree.evictable_size(), tree.protected_size(), tree.total_size()) == (op["evictable"], op["protected"], op["total"]), (page, i, op)


def test_reference_main_scenario(golden_dir):
    # radix_cache.py:849-863
    g = json.loads((golden_dir / "radix_trace.json").read_text())["main_scenario"]
    tree = RadixCache.create_simulated()
    for ids in ([1, 2, 3], [1, 2, 3], [1, 2, 4, 5], [1, 2, 4, 5, 6, 7], [8, 9, 10, 11, 12]):
        tree.insert(InsertParams(key=_key(ids)))
    m = tree.match_prefix(MatchPrefixParams(key=_key([1, 2, 3, 13, 14])))
    assert m.device_indices.tolist() == g["match"] == [1, 2, 3]
    assert tree.total_size() == g["total"]


def test_radix_key_semantics():
    # test/registered/unit/mem_cache/test_radix_cache_unit.py:210-258 (match lengths incl. page rounding)
    a, b = _key([1, 2, 3, 4, 5, 6, 7]), _key([1, 2, 3, 4, 9, 9])
    assert a.match(b) == 4
    assert a.match(b, page_size=2) == 4
    assert a.match(b, page_size=3) == 3
    assert a.match(_key([]), page_size=1) == 0
    assert a.match(a) == 7 and a.match(a, page_size=4) == 4
    assert len(a.page_aligned(4)) == 4 and len(a.page_aligned(8)) == 0
    assert a[2:5].token_ids.tolist() == [3, 4, 5]
    assert a.child_key(1) == 1 and a.child_key(3) == (1, 2, 3)
    assert RadixKey(array("q", [5]), extra_key="lora").child_key(1) == ("lora", 5)
    with pytest.raises(ValueError):
        a.match(RadixKey(array("q", [1]), extra_key="x"))


def test_extra_key_namespaces_are_disjoint():
    tree = RadixCache.create_simulated()
    tree.insert(InsertParams(key=RadixKey(array("q", [1, 2, 3]), extra_key="a"), value=torch.tensor([10, 11, 12])))
    tree.insert(InsertParams(key=RadixKey(array("q", [1, 2, 3]), extra_key="b"), value=torch.tensor([20, 21, 22])))
    assert tree.match_prefix(RadixKey(array("q", [1, 2, 3]), extra_key="a")).device_indices.tolist() == [10, 11, 12]
    assert tree.match_prefix(RadixKey(array("q", [1, 2, 3]), extra_key="b")).device_indices.tolist() == [20, 21, 22]
    assert tree.match_prefix(_key([1, 2, 3])).device_indices.tolist() == []


@pytest.mark.parametrize("page", [1, 2, 16])
def test_against_bruteforce_oracle(page):
    rnd = random.Random(page)
    tree = RadixCache.create_simulated(mock_allocator=mock.Mock(device="cpu"), page_size=page)
    model = BruteForcePrefixCache(page_size=page)
    slot = 1
    for _ in range(400):
        ids = [rnd.randrange(4) for _ in range(rnd.randint(0, 70))]
        if rnd.random() < 0.5 and ids:
            vals = list(range(slot, slot + len(ids)))
            slot += len(ids)
            got = tree.insert(InsertParams(key=_key(ids), value=torch.tensor(vals, dtype=torch.int64))).prefix_len
            assert got == model.insert(ids, vals)
        else:
            assert tree.match_prefix(_key(ids)).device_indices.tolist() == model.match(ids)


def test_lock_protects_from_eviction_and_disable():
    alloc = mock.Mock(device="cpu")
    tree = RadixCache.create_simulated(mock_allocator=alloc)
    tree.insert(InsertParams(key=_key([1, 2, 3, 4]), value=torch.tensor([1, 2, 3, 4])))
    tree.insert(InsertParams(key=_key([1, 2, 9]), value=torch.tensor([1, 2, 9])))
    m = tree.match_prefix(_key([1, 2, 3, 4]))
    tree.inc_lock_ref(m.last_device_node)
    assert tree.protected_size() == 4 and tree.evictable_size() == 1
    assert tree.evict(EvictParams(100)).num_tokens_evicted == 1          # only the [9] leaf can go
    tree.dec_lock_ref(m.last_device_node)
    assert tree.evict(EvictParams(100)).num_tokens_evicted == 4
    assert tree.total_size() == 0
    off = RadixCache.create_simulated(disable=True)
    assert off.insert(InsertParams(key=_key([1]))).prefix_len == 0
    assert off.match_prefix(_key([1])).device_indices.numel() == 0


def test_eviction_policies():
    for policy, first in (("lru", [1]), ("mru", [3]), ("fifo", [1]), ("filo", [3])):
        alloc = mock.Mock(device="cpu")
        tree = RadixCache(None, alloc, eviction_policy=policy)
        for t in (1, 2, 3):
            tree.insert(InsertParams(key=_key([t]), value=torch.tensor([t])))
        tree.evict(EvictParams(1))
        assert alloc.free_segment.call_args_list[0].args[0].tolist() == first, policy


def test_token_allocator_matches_reference_semantics():
    # allocator/token.py:40-76: ascending slots starting at 1, free appends, need_sort merges
    a = TokenToKVPoolAllocator(16, torch.bfloat16, "cpu")
    assert a.available_size() == 16
    x = a.alloc(5)
    assert x.tolist() == [1, 2, 3, 4, 5]
    assert a.alloc(12) is None
    a.free(x[1:3])
    assert a.available_size() == 13
    assert a.alloc(11).tolist() == list(range(6, 17))
    assert a.alloc(2).tolist() == [2, 3]
    s = TokenToKVPoolAllocator(8, torch.bfloat16, "cpu", need_sort=True)
    y = s.alloc(8)
    s.free(y[5:]); s.free(y[:2])
    assert s.alloc(4).tolist() == [1, 2, 6, 7]
    a.free_group_begin(); a.free(torch.tensor([9])); a.free(torch.tensor([8])); a.free_group_end()
    assert sorted(a.free_pages.tolist()) == [8, 9]


def test_paged_allocator_free_segments():
    # base.py:133-149 boundary rule + paged.py:281-313 strided representatives
    p = PagedTokenToKVPoolAllocator(64, 4, torch.bfloat16, "cpu")
    idx = p.alloc(16)
    assert idx.tolist() == list(range(4, 20)) and p.available_size() == 48
    row = idx                      # one request's kv row: pages 1,2,3,4
    p.free_segments([(row[2:6], 2), (row[6:16], 6)])   # second segment starts inside page 2
    assert sorted(p.free_pages.tolist()[:4]) == [1, 2, 3, 4]
    assert p.available_size() == 64


def _drive_prefill_like(fast: bool, page: int, seed: int):
    """Two prefill rounds + finishes over CPU pools, the way harness/engine.py drives the cache."""
    import random

    from sglang_amd.harness.engine import Req
    from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator
    from sglang_amd.mem_cache.memory_pool import ReqToTokenPool
    from sglang_amd.mem_cache.radix_cache import EvictParams, MatchPrefixParams, RadixCache, RadixKey

    dev = torch.device("cpu")
    B, ctx, size = 12, 96, 12 * 96
    r2t = ReqToTokenPool(B, ctx, dev)
    alloc = (TokenToKVPoolAllocator(size, torch.bfloat16, dev, None) if page == 1
             else PagedTokenToKVPoolAllocator(size, page, torch.bfloat16, dev, None))
    tree = RadixCache(r2t, alloc, page)
    tree.fast_unfinished_path = fast
    rnd = random.Random(seed)
    shared = [[rnd.randrange(50) for _ in range(40)] for _ in range(2)]
    prompts = [shared[i % 2] + [rnd.randrange(50) for _ in range(rnd.randrange(1, 30))] for i in range(8)]
    prompts += [list(prompts[0]), list(prompts[1][:45])]          # a duplicate prompt and a strict prefix of another
    reqs = [Req(i, p, 4) for i, p in enumerate(prompts)]
    log = []

    def prefill(rs):
        for q in rs:
            key = RadixKey(q.origin_input_ids[: len(q.origin_input_ids) - 1], q.extra_key, q.cache_salt)
            m = tree.match_prefix(MatchPrefixParams(key=key))
            q.prefix_indices, q.last_node = m.device_indices, m.last_device_node
            q.cached_tokens = int(m.device_indices.numel())
            q.cache_protected_len = q.cached_tokens
            tree.inc_lock_ref(q.last_node)
        assert r2t.alloc(rs) is not None
        for q in rs:
            n = len(q.origin_input_ids)
            if page == 1:
                loc = alloc.alloc(n - q.cached_tokens)
            else:
                # the cached prefix ends on a page boundary, so the new tokens start a fresh page (plain alloc of
                # whole pages; alloc_extend is a HIP kernel and this test has no GPU)
                need = (n - q.cached_tokens + page - 1) // page
                loc = alloc.alloc(need * page)[: n - q.cached_tokens]
            assert loc is not None
            r2t.req_to_token[q.req_pool_idx, : q.cached_tokens] = q.prefix_indices.to(torch.int32)
            r2t.req_to_token[q.req_pool_idx, q.cached_tokens: n] = loc.to(torch.int32)
        for q in rs:
            tree.cache_unfinished_req(q)
            log.append((q.rid, q.cache_protected_len, q.prefix_indices.tolist(),
                        r2t.req_to_token[q.req_pool_idx, : len(q.origin_input_ids)].tolist()))

    prefill(reqs[:2])
    prefill(reqs[2:])
    log.append(("sizes", tree.evictable_size(), tree.protected_size(), tree.total_size(), alloc.available_size()))
    for q in reqs[:5]:
        tree.cache_finished_req(q, kv_len_to_handle=len(q.origin_input_ids))
        r2t.free(q)
    log.append(("sizes", tree.evictable_size(), tree.protected_size(), tree.total_size(), alloc.available_size()))
    before = alloc.available_size()
    tree.evict(EvictParams(num_tokens=60))                          # eviction order depends on the recency stamps
    log.append(("evicted", alloc.available_size() - before, tree.evictable_size(), sorted(tree.all_values_flatten().tolist())))
    return log


@pytest.mark.parametrize("page", [1, 4])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_unfinished_req_fast_path_equals_reference_two_pass_form(page, seed):
    """cache_unfinished_req's shortcut (no second match_prefix / row rewrite when the tree took the request's own
    slots) leaves exactly the state of the reference's form (radix_cache.py:516-584): rows, locks, sizes, and
    the eviction order that follows from the recency stamps."""
    assert _drive_prefill_like(True, page, seed) == _drive_prefill_like(False, page, seed)
