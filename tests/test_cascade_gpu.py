"""GPU parity tests of the shared-prefix (cascade) decode attention: the device plan is checked against a
host restatement, and the attention output against the CPU oracle
(oracle/ops.py decode_attention = torch_native_backend.py:176-277) and the plain paged decode kernel."""
import pytest
import torch

from oracle import ops as oo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _k():
    from sglang_amd import kernels

    return kernels


def _batch(device, lens, groups, shared, Hq, Hkv, D, seed=0, extra_rows=8):
    """lens[b] kv length; groups[b] = group id or -1; shared[g] = shared prefix length of group g."""
    g = torch.Generator().manual_seed(seed)
    B = len(lens)
    ctx = max(lens) + extra_rows
    slots = sum(lens) + 64
    kc = (torch.randn((slots, Hkv, D), generator=g) * 0.5).to(BF)
    vc = (torch.randn((slots, Hkv, D), generator=g) * 0.5).to(BF)
    perm = (torch.randperm(slots - 1, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros((B + 1, ctx), dtype=torch.int32)
    off = 0
    first_of_group = {}
    for b in range(B):
        r2t[b + 1, :lens[b]] = perm[off: off + lens[b]]
        off += lens[b]
        gi = groups[b]
        if gi >= 0:
            if gi in first_of_group:
                n = min(shared[gi], lens[b] - 1)
                r2t[b + 1, :n] = r2t[first_of_group[gi] + 1, :n]
            else:
                first_of_group[gi] = b
    q = (torch.randn((B, Hq, D), generator=g) * 0.5).to(BF)
    pool = torch.arange(1, B + 1)
    seq = torch.tensor(lens, dtype=torch.int32)
    return q, kc, vc, r2t, pool, seq


def _run(device, q, kc, vc, r2t, pool, seq, Hq, Hkv, D, min_shared=128):
    K = _k()
    B = q.shape[0]
    ws = K.CascadeWorkspace(B, Hq, D, r2t.shape[1], device)
    r2t_d, pool_d, seq_d = r2t.to(device), pool.to(device), seq.to(device)
    K.cascade_plan(ws, r2t_d, pool_d, seq_d, Hq, Hkv, min_shared)
    out = torch.empty((B, Hq, D), dtype=BF, device=device)
    K.cascade_decode_attention(ws, q.to(device), kc.to(device), vc.to(device), out, r2t_d, pool_d, seq_d, D ** -0.5)
    return out.cpu(), K.cascade_plan_summary(ws, B)


def _check(out, q, kc, vc, r2t, pool, seq, D):
    ref = oo.decode_attention(q, kc, vc, r2t, pool, seq.long(), D ** -0.5, compute_dtype=torch.float32)
    err = (out.float() - ref.float()).abs()
    assert float(err.max()) <= 2.0 ** -8 * float(ref.float().abs().max()) + 2e-3, float(err.max())


def test_plan_groups_the_bench_pattern(device):
    """4 groups x 16 requests (leaders first, like the engine's running list), 896 shared tokens."""
    Hq, Hkv, D = 32, 8, 128
    B, P = 64, 16
    order = [g * P for g in range(4)] + [g * P + i for g in range(4) for i in range(1, P)]
    groups = [o // P for o in order]
    lens = [1030 + (i % 7) for i in range(B)]
    q, kc, vc, r2t, pool, seq = _batch(device, lens, groups, {g: 896 for g in range(4)}, Hq, Hkv, D)
    out, plan = _run(device, q, kc, vc, r2t, pool, seq, Hq, Hkv, D)
    assert plan["n_groups"] == 4 and plan["group_kvlen"] == [896] * 4
    assert plan["group_qo"] == [0, 16, 32, 48, 64]
    assert sorted(plan["member_rows"]) == list(range(B))
    for gi in range(4):
        rows = plan["member_rows"][16 * gi: 16 * gi + 16]
        assert rows == sorted(rows) and {groups[r] for r in rows} == {gi}
    assert plan["req_shared"] == [896] * B
    CH = _k().native.lib().sgl_amd_cascade_chunk_tokens()
    ns = (896 + CH - 1) // CH
    assert plan["n_shared_items"] == 4 * ns and len(set(plan["items"])) == 4 * ns     # 16 members x 4 heads = one 64-row item
    # private part: tokens [896, len) of every request in CH-token items, slots behind the shared ones
    want = sorted((b, ns + j, 896 + CH * j, min(CH, lens[b] - 896 - CH * j)) for b in range(B)
                  for j in range((lens[b] - 896 + CH - 1) // CH))
    assert sorted(plan["private_items"]) == want and plan["n_items"] == 4 * ns + len(want)
    _check(out, q, kc, vc, r2t, pool, seq, D)


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (14, 2, 64), (16, 4, 128), (4, 4, 64)])
def test_cascade_matches_oracle_mixed_batch(device, Hq, Hkv, D):
    """Groups of different sizes / shared lengths, singletons, a too-short share, ragged suffixes,
    a group whose members share DIFFERENT lengths with the leader (min wins), 40 members (several row tiles)."""
    lens, groups, shared = [], [], {}
    def add(n, gid, ln):
        for i in range(n):
            lens.append(ln + 3 * i)
            groups.append(gid)
    add(3, 0, 700); shared[0] = 512
    add(1, -1, 333)
    add(40, 1, 300); shared[1] = 200          # rounds down to 192
    add(2, 2, 150); shared[2] = 70            # below min_shared: no group
    add(1, -1, 1)                             # a request with a single token
    add(2, 3, 900); shared[3] = 899
    q, kc, vc, r2t, pool, seq = _batch(device, lens, groups, shared, Hq, Hkv, D, seed=Hq + D)
    # make one member of group 0 diverge earlier than the others
    r2t[3, 300:512] = r2t[4, 600:812]
    out, plan = _run(device, q, kc, vc, r2t, pool, seq, Hq, Hkv, D)
    assert plan["n_groups"] == 3
    assert plan["group_kvlen"] == [256, 192, 896]          # 300 -> 256, 200 -> 192, 899 -> 896
    mpi = 64 // (Hq // Hkv)                     # members per item: 64 (member, head) rows per workgroup
    tiles = lambda members: (members + mpi - 1) // mpi
    CH = _k().native.lib().sgl_amd_cascade_chunk_tokens()
    ch = lambda kv: (kv + CH - 1) // CH
    assert plan["n_shared_items"] == ch(256) * tiles(3) + ch(192) * tiles(40) + ch(896) * tiles(2)
    _check(out, q, kc, vc, r2t, pool, seq, D)


def test_cascade_without_sharing_equals_plain_decode(device):
    K = _k()
    Hq, Hkv, D = 32, 8, 128
    lens = [517, 64, 1, 129, 1000, 33, 257]
    q, kc, vc, r2t, pool, seq = _batch(device, lens, [-1] * len(lens), {}, Hq, Hkv, D, seed=3)
    out, plan = _run(device, q, kc, vc, r2t, pool, seq, Hq, Hkv, D)
    assert plan["n_groups"] == 0 and plan["n_shared_items"] == 0 and plan["req_shared"] == [0] * len(lens)
    CH = K.native.lib().sgl_amd_cascade_chunk_tokens()
    assert plan["n_items"] == sum((ln + CH - 1) // CH for ln in lens)
    _check(out, q, kc, vc, r2t, pool, seq, D)
    plain = torch.empty_like(q, device=device)
    K.decode_attention(q.to(device), kc.to(device), vc.to(device), plain, r2t.to(device), pool.to(device), seq.to(device),
                       D ** -0.5)
    assert float((plain.cpu().float() - out.float()).abs().max()) <= 2.0 ** -7


def test_cascade_plan_replans_every_step(device):
    """The same workspace is reused across steps with different groupings (stale slots must not leak)."""
    K = _k()
    Hq, Hkv, D = 16, 4, 128
    B = 12
    ws = None
    for step, (groups, shared) in enumerate([([0] * 6 + [1] * 6, {0: 640, 1: 384}), ([-1] * 12, {}),
                                             ([0] * 12, {0: 256})]):
        lens = [700 + 5 * i for i in range(B)]
        q, kc, vc, r2t, pool, seq = _batch(device, lens, groups, shared, Hq, Hkv, D, seed=step)
        if ws is None:
            ws = K.CascadeWorkspace(B, Hq, D, r2t.shape[1] + 64, device)
        r2t_d, pool_d, seq_d = r2t.to(device), pool.to(device), seq.to(device)
        K.cascade_plan(ws, r2t_d, pool_d, seq_d, Hq, Hkv)
        out = torch.empty((B, Hq, D), dtype=BF, device=device)
        K.cascade_decode_attention(ws, q.to(device), kc.to(device), vc.to(device), out, r2t_d, pool_d, seq_d, D ** -0.5)
        _check(out.cpu(), q, kc, vc, r2t, pool, seq, D)


@pytest.mark.parametrize("where", ["shared", "private"])
def test_cascade_and_plain_decode_survive_a_dominating_key(device, where):
    """Scores 250 log2-units apart inside one request (a query aligned with one huge key): every partial maximum has to
    reach every lane and every merge, or exp2 overflows (the extend kernel's lane-row reduction once did not:
    test_kernels_gpu.py every_workgroup_shape).  The key sits in the shared prefix or in a private suffix, at a position
    that is not a multiple of 4 or 16."""
    Hq, Hkv, D = 32, 8, 128
    lens, groups = [700, 705, 650, 333], [0, 0, 0, -1]
    q, kc, vc, r2t, pool, seq = _batch(device, lens, groups, {0: 512}, Hq, Hkv, D, seed=5)
    b, pos = 1, (301 if where == "shared" else 641)
    slot = int(r2t[int(pool[b]), pos])
    base = (torch.randn(D, generator=torch.Generator().manual_seed(9)) * 0.6).to(BF)
    kc[slot] = (base * 8).to(BF)                  # all kv heads of that token
    q[b] = (base * 7).to(BF)                      # all q heads of request b: score ~ 56 |base|^2 / sqrt(D) ~ 250 log2-units
    out, plan = _run(device, q, kc, vc, r2t, pool, seq, Hq, Hkv, D)
    assert plan["n_groups"] == 1
    _check(out, q, kc, vc, r2t, pool, seq, D)
    # the plain paged kernel on the same batch, one and several KV splits
    K = _k()
    for splits in (1, 4):
        ws = K.decode_workspace(len(lens), Hq, D, splits, device) if splits > 1 else (None, None)
        o = torch.empty((len(lens), Hq, D), dtype=BF, device=device)
        K.decode_attention(q.to(device), kc.to(device), vc.to(device), o, r2t.to(device), pool.to(device), seq.to(device), D ** -0.5,
                           splits, ws[0], ws[1])
        assert not bool(torch.isnan(o.float()).any())
        _check(o.cpu(), q, kc, vc, r2t, pool, seq, D)


@pytest.mark.parametrize("seed", range(6))
def test_device_plan_equals_host_restatement_on_random_batches(device, seed):
    """The plan kernel (parallel grouping passes, one wave accepting the groups, rank-by-counting member order) against
    oracle/host.py cascade_plan on random batches: random group sizes and shared lengths, members diverging early,
    singletons, one-token requests, interleaved batch order, up to 200 requests."""
    import random

    from oracle import host as oh

    rnd = random.Random(seed)
    Hq, Hkv, D = 32, 8, 128
    B = rnd.choice([5, 37, 64, 130, 200])
    n_groups = rnd.randint(1, max(1, B // 3))
    lens, groups, shared = [], [], {}
    for b in range(B):
        gi = rnd.randint(-1, n_groups - 1) if rnd.random() < 0.8 else -1
        groups.append(gi)
        lens.append(rnd.choice([1, 2, 65, 129]) if rnd.random() < 0.1 else rnd.randint(130, 900))
    for gi in range(n_groups):
        shared[gi] = rnd.choice([40, 100, 128, 200, 384, 512, 700])
    q, kc, vc, r2t, pool, seq = _batch(device, lens, groups, shared, Hq, Hkv, D, seed=seed)
    for _ in range(3):                                    # some members leave the shared prefix early
        b = rnd.randrange(B)
        if lens[b] > 140:
            cut = rnd.randint(1, lens[b] - 2)
            r2t[b + 1, cut] = 0
    K = _k()
    ws = K.CascadeWorkspace(B, Hq, D, r2t.shape[1], device)
    K.cascade_plan(ws, r2t.to(device), pool.to(device), seq.to(device), Hq, Hkv)
    got = K.cascade_plan_summary(ws, B)
    want = oh.cascade_plan(r2t.numpy(), pool.tolist(), lens, Hq // Hkv, chunk=K.native.lib().sgl_amd_cascade_chunk_tokens(),
                           max_context_len=r2t.shape[1], max_items=ws.max_items)
    assert got["req_shared"] == want["req_shared"]
    assert got["n_groups"] == len(want["groups"])
    assert got["group_kvlen"] == [kv for _, kv, _ in want["groups"]]
    assert got["member_rows"] == want["member_rows"]
    assert got["items"] == want["shared_items"]
    assert sorted(got["private_items"]) == sorted(want["private_items"])
    assert got["n_items"] == len(want["shared_items"]) + len(want["private_items"])
    grouped = set(want["member_rows"])
    assert got["batch_order"] == want["member_rows"] + [b for b in range(B) if b not in grouped]
