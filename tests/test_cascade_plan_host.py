"""The shared-prefix decode plan, restated on the host (oracle/host.py cascade_plan): its invariants on random
batches, and the layouts the GPU tests assert on the device plan (tests/test_cascade_gpu.py) reproduced here."""
import random

import numpy as np
import pytest

from oracle import host as oh
from sglang_amd import native

CH = native.lib().sgl_amd_cascade_chunk_tokens()      # kv tokens per plan item (the library loads without a GPU)


def _batch(rnd, B, groups, ctx):
    """req_to_token rows: request b belongs to group g (or -1); members copy a prefix of their group's first row."""
    r2t = np.zeros((B + 1, ctx), dtype=np.int32)
    nxt = 1
    lens, first_of = [], {}
    for b in range(B):
        ln = rnd.randrange(2, ctx - 1)
        lens.append(ln)
        r2t[b + 1, :ln] = np.arange(nxt, nxt + ln)
        nxt += ln
        g = groups[b]
        if g >= 0:
            if g in first_of:
                src = first_of[g]
                n = min(rnd.randrange(1, ctx), ln - 1, lens[src] - 1)
                r2t[b + 1, :n] = r2t[src + 1, :n]
            else:
                first_of[g] = b
    return r2t, list(range(1, B + 1)), lens


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("group", [1, 4, 7, 16])
def test_every_token_is_covered_exactly_once(seed, group):
    rnd = random.Random(seed)
    B, ctx = rnd.randrange(1, 40), 700
    groups = [rnd.choice([-1, 0, 1, 2, 3]) for _ in range(B)]
    r2t, pool, lens = _batch(rnd, B, groups, ctx)
    plan = oh.cascade_plan(r2t, pool, lens, group, chunk=CH, max_context_len=ctx)
    mpi = 64 // group
    covered = [np.zeros(n, dtype=np.int32) for n in lens]
    slots = [set() for _ in range(B)]
    for gi, slot, first, members in plan["shared_items"]:
        leader, kv, mem = plan["groups"][gi]
        assert 1 <= members <= mpi and kv % 64 == 0 and kv >= 128
        lo, hi = slot * CH, min(kv, slot * CH + CH)
        for m in plan["member_rows"][first: first + members]:
            assert m in mem
            # the shared rows really are the same pool slots for every member
            assert np.array_equal(r2t[pool[m]][lo:hi], r2t[pool[leader]][lo:hi])
            covered[m][lo:hi] += 1
            slots[m].add(slot)
    for b, slot, kv_begin, kv_n in plan["private_items"]:
        assert 1 <= kv_n <= CH and kv_begin >= plan["req_shared"][b]
        covered[b][kv_begin: kv_begin + kv_n] += 1
        assert slot not in slots[b]
        slots[b].add(slot)
    # the chunk kernel's one-workgroup-per-unit grid is sized for batch x (chunks + 1) items (cascade_attention.hip): a request
    # never owns more than that many, however the batch is grouped
    chunks = (ctx + CH - 1) // CH
    assert len(plan["shared_items"]) + len(plan["private_items"]) <= B * (chunks + 1)
    for b in range(B):
        assert (covered[b] == 1).all(), f"request {b}"
        n = (plan["req_shared"][b] + CH - 1) // CH + ((lens[b] - plan["req_shared"][b] + CH - 1) // CH if lens[b] > plan["req_shared"][b] else 0)
        assert slots[b] == set(range(n))                     # the merge kernel reads slots [0, n)
        assert plan["req_shared"][b] < lens[b]               # the newest token is never in a shared part


def test_bench_pattern_layout():
    """4 groups x 16 requests sharing 896 tokens (leaders first), lengths 1030..1036: the numbers
    tests/test_cascade_gpu.py::test_plan_groups_the_bench_pattern asserts on the device plan."""
    B, P = 64, 16
    order = [g * P for g in range(4)] + [g * P + i for g in range(4) for i in range(1, P)]
    grp = [o // P for o in order]
    lens = [1030 + (i % 7) for i in range(B)]
    r2t = np.zeros((B + 1, 1100), dtype=np.int32)
    nxt, first_of = 1, {}
    for b in range(B):
        r2t[b + 1, :lens[b]] = np.arange(nxt, nxt + lens[b])
        nxt += lens[b]
        if grp[b] in first_of:
            r2t[b + 1, :896] = r2t[first_of[grp[b]] + 1, :896]
        else:
            first_of[grp[b]] = b
    plan = oh.cascade_plan(r2t, list(range(1, B + 1)), lens, 4, chunk=CH, max_context_len=1100)
    assert [kv for _, kv, _ in plan["groups"]] == [896] * 4 and plan["req_shared"] == [896] * B
    ns = (896 + CH - 1) // CH                           # shared items per group (one member tile: 16 x 4 rows)
    assert len(plan["shared_items"]) == 4 * ns and all(m == 16 for *_, m in plan["shared_items"])
    want = sorted((b, ns + j, 896 + CH * j, min(CH, lens[b] - 896 - CH * j)) for b in range(B)
                  for j in range((lens[b] - 896 + CH - 1) // CH))
    assert sorted(plan["private_items"]) == want


def test_no_sharing_means_private_chunks_only():
    lens = [517, 64, 1, 129, 1000, 33, 257]
    r2t = np.zeros((len(lens) + 1, 1024), dtype=np.int32)
    nxt = 1
    for b, n in enumerate(lens):
        r2t[b + 1, :n] = np.arange(nxt, nxt + n)
        nxt += n
    plan = oh.cascade_plan(r2t, list(range(1, len(lens) + 1)), lens, 4, chunk=CH, max_context_len=1024)
    assert plan["groups"] == [] and plan["req_shared"] == [0] * len(lens)
    assert len(plan["private_items"]) == sum((n + CH - 1) // CH for n in lens)


@pytest.mark.parametrize("group,members", [(1, 2), (1, 64), (4, 2), (4, 16), (4, 17), (4, 33), (16, 5), (64, 3), (64, 40)])
def test_item_count_never_exceeds_the_grid_bound_for_adversarial_groupings(group, members):
    """The grid bound of the one-workgroup-per-unit chunk kernel, batch x (ceil(width / chunk) + 1), on the groupings that maximise
    shared items per member: groups of `members` requests (member tiles round UP) whose shared parts end at every offset around a
    chunk boundary, lengths up to the table's width."""
    width = 1160
    chunks = (width + CH - 1) // CH
    for shared in (128, 191, 192, 255, 256, 320, 896, 1087):
        for tail in (1, 63, 64, 65, 127, 128, 129):
            B = 2 * members + 3
            lens = [min(width - 1, shared + tail + (b % 5)) for b in range(B)]
            r2t = np.zeros((B + 1, width), dtype=np.int32)
            nxt = 1
            for b in range(B):
                r2t[b + 1, :lens[b]] = np.arange(nxt, nxt + lens[b])
                nxt += lens[b]
            for g in range(2):                                  # two groups of `members`, three loners
                lead = g * members
                n = min(shared, min(lens[lead: lead + members]) - 1)
                for b in range(lead + 1, lead + members):
                    r2t[b + 1, :n] = r2t[lead + 1, :n]
            plan = oh.cascade_plan(r2t, list(range(1, B + 1)), lens, group, chunk=CH, max_context_len=width)
            n_items = len(plan["shared_items"]) + len(plan["private_items"])
            assert n_items <= B * (chunks + 1), (shared, tail, n_items, B * (chunks + 1))
