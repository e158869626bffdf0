"""TP > 1 through the plug-in's hooks, with the real kernels: two processes sharing GPU 0 each build a stand-in of the
reference's GroupCoordinator (srt/distributed/parallel_state.py:237-420 -- the attribute names its constructor binds, a
gloo `cpu_group`, a `device_group`; RCCL refuses two ranks on one device, so that one is gloo too) wrapped by the four
AROUND hooks exactly as HookRegistry.apply_hooks wraps them (hook(original, self, ...)).

  * constructing the group attaches an XgmiAllReduce (IPC handles over the group's own cpu_group, the start-up self-test
    against the group's own device_group);
  * `all_reduce` of decode- and prefill-sized bf16 messages runs the xGMI kernels (the reference's method is NOT entered)
    and returns the fp32 rank-order sum rounded once; fp32 tensors and oversized messages reach the reference's method;
  * `fused_allreduce_rmsnorm` returns (RMSNorm(sum + residual), residual) with the residual updated in place, equal to
    the oracle's fused_add_rmsnorm of the summed input;
  * `all_gather` along the last dimension equals the concatenation of the ranks' shards;
  * a hipGraph captured around the hooked all_reduce replays correctly;
  * the fused decode layer of fused_decode.py at TP = 2 (decode_model with the group's communicator) equals the
    operator-by-operator TP = 2 layer loop that all-reduces through the hooked group."""
import os
import sys
import traceback
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _make_group_class(dist, calls):
    """The reference's GroupCoordinator, restated as far as the hooks touch it, then wrapped like apply_hooks does."""
    import functools

    from sglang_amd import tp_hooks

    class GroupCoordinator:
        def __init__(self, group_ranks, local_rank, torch_distributed_backend, use_pynccl, use_pymscclpp, use_custom_allreduce,
                     use_torch_symm_mem_all_reduce, use_hpu_communicator, use_xpu_communicator, use_npu_communicator,
                     use_message_queue_broadcaster=False, group_name=None):
            self.unique_name = f"{group_name or 'anonymous'}:0"
            self.rank = dist.get_rank()
            self.local_rank = local_rank
            for ranks in group_ranks:                                        # parallel_state.py:345-410
                device_group = dist.new_group(ranks, backend=torch_distributed_backend)
                cpu_group = dist.new_group(ranks, backend="gloo")
                if self.rank in ranks:
                    self.ranks, self.world_size, self.rank_in_group = ranks, len(ranks), ranks.index(self.rank)
                    self.device_group, self.cpu_group = device_group, cpu_group
            self.device = torch.device("cuda", local_rank)
            self.use_custom_allreduce = use_custom_allreduce
            self.ca_comm = None

        def all_reduce(self, input_):                                        # :648-758, the plain torch.distributed leg
            calls.append("all_reduce")
            dist.all_reduce(input_, group=self.device_group)
            return input_

        def fused_allreduce_rmsnorm(self, input_, residual_inp_, weight_, eps):   # :774-833 with ca_comm None
            calls.append("fused_allreduce_rmsnorm")
            return None

        def all_gather(self, input_, dim=-1, output_tensor_list=None):       # :1273-1345
            calls.append("all_gather")
            parts = [torch.empty_like(input_) for _ in range(self.world_size)]
            dist.all_gather(parts, input_.contiguous(), group=self.device_group)
            return torch.cat(parts, dim=dim)

    for target, hook in zip(tp_hooks.HOOK_TARGETS, tp_hooks._HOOKS):
        name = target.rsplit(".", 1)[1]
        orig = getattr(GroupCoordinator, name)

        def wrapper(*args, __orig=orig, __hook=hook, **kwargs):             # hook_registry.py _wrap_fn, HookType.AROUND
            return __hook(__orig, *args, **kwargs)

        setattr(GroupCoordinator, name, functools.wraps(orig)(wrapper))
    return GroupCoordinator


def _inputs(world, rows, hidden, seed):
    xs = []
    for r in range(world):
        g = torch.Generator(device="cpu").manual_seed(1000 * seed + r)
        xs.append((torch.randn((rows, hidden), generator=g) * 0.5).to(torch.bfloat16))
    return xs


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, str(ROOT))
        os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                           "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        import torch.distributed as dist

        from oracle import ops as oo
        from sglang_amd import tp_hooks

        torch.cuda.set_device(0)
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        calls = []
        GC = _make_group_class(dist, calls)
        grp = GC([list(range(world))], 0, "gloo", False, False, True, False, False, False, False, group_name="tp")
        comm = tp_hooks.communicator_of(grp)
        assert comm is not None, "no communicator was attached to a 2-rank GPU group built with use_custom_allreduce"
        # a second group over the same ranks (attention_tp of a plain TP launch) gets a communicator of its OWN -- workspace and
        # flag counters -- as the reference gives every group its own ca_comm (parallel_state.py:405-470); both work, interleaved
        grp2 = GC([list(range(world))], 0, "gloo", False, False, True, False, False, False, False, group_name="attention_tp")
        comm2 = tp_hooks.communicator_of(grp2)
        assert comm2 is not None and comm2 is not comm and comm2._own != comm._own
        for i in range(3):
            xs = _inputs(world, 64, 4096, 40 + i)
            want = torch.stack([x.float() for x in xs]).sum(0).to(torch.bfloat16)
            s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            torch.cuda.synchronize()
            with torch.cuda.stream(s1):
                got1 = grp.all_reduce(xs[rank].to(dev).clone())
            with torch.cuda.stream(s2):
                got2 = grp2.all_reduce(xs[rank].to(dev).clone())
            torch.cuda.synchronize()
            assert torch.equal(got1.cpu(), want) and torch.equal(got2.cpu(), want), i
        # a group built WITHOUT the hint keeps the reference's collectives
        plain = GC([list(range(world))], 0, "gloo", False, False, False, False, False, False, False, group_name="tp")
        assert tp_hooks.communicator_of(plain) is None

        # ---- all_reduce: decode-sized (one-shot) and prefill-sized (two-stage) bf16 messages ---------------------
        for i, (rows, hidden) in enumerate([(64, 4096), (1, 8), (64, 8192), (1100, 1024), (2048, 4096)]):
            xs = _inputs(world, rows, hidden, i)
            want = torch.stack([x.float() for x in xs]).sum(0).to(torch.bfloat16)
            n = len(calls)
            got = grp.all_reduce(xs[rank].to(dev).clone())
            assert len(calls) == n, "the reference's all_reduce was entered for a message the kernels take"
            assert torch.equal(got.cpu(), want), (rows, hidden)
        f32 = torch.full((8, 16), float(rank + 1), device=dev)
        out = grp.all_reduce(f32)
        assert calls[-1] == "all_reduce" and torch.equal(out.cpu(), torch.full((8, 16), float(sum(range(1, world + 1)))))
        assert plain.all_reduce(torch.ones((4, 8), dtype=torch.bfloat16, device=dev)).float().mean().item() == world and calls[-1] == "all_reduce"

        # ---- fused_allreduce_rmsnorm: (normed, residual), residual in place ---------------------------------------
        for i, (rows, hidden) in enumerate([(64, 4096), (5, 8192), (512, 4096)]):
            xs = _inputs(world, rows, hidden, 10 + i)
            g = torch.Generator().manual_seed(77 + i)
            res0 = torch.randn((rows, hidden), generator=g).to(torch.bfloat16)
            w = (1.0 + 0.1 * torch.randn(hidden, generator=g)).to(torch.bfloat16)
            summed = torch.stack([x.float() for x in xs]).sum(0).to(torch.bfloat16)
            want_out, want_res = oo.fused_add_rmsnorm(summed, res0, w, 1e-5)
            res = res0.to(dev).clone()
            n = len(calls)
            got = grp.fused_allreduce_rmsnorm(xs[rank].to(dev), res, w.to(dev), 1e-5)
            assert len(calls) == n and isinstance(got, tuple) and got[1] is res
            assert torch.equal(res.cpu(), want_res), "residual"
            err = (got[0].cpu().float() - want_out.float()).abs()
            assert float(err.max()) <= 2.0 ** -7 * float(want_out.float().abs().max()) and float((err > 0).float().mean()) < 0.01
        assert plain.fused_allreduce_rmsnorm(xs[rank].to(dev), res, w.to(dev), 1e-5) is None    # the reference's "no fused path"

        # ---- all_gather along the last dimension --------------------------------------------------------------------
        for i, (rows, cols) in enumerate([(64, 16032), (7, 1000)]):
            xs = _inputs(world, rows, cols, 60 + i)
            n = len(calls)
            got = grp.all_gather(xs[rank].to(dev))
            assert len(calls) == n and torch.equal(got.cpu(), torch.cat(xs, dim=1))

        # ---- the hooked all_reduce inside a hipGraph ------------------------------------------------------------------
        static_in = torch.zeros((64, 4096), dtype=torch.bfloat16, device=dev)
        holder = []
        grp.all_reduce(static_in.clone())
        torch.cuda.synchronize(); dist.barrier()
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(graph, stream=s):
                holder.append(grp.all_reduce(static_in))
        torch.cuda.current_stream().wait_stream(s)
        for i in range(3):
            xs = _inputs(world, 64, 4096, 80 + i)
            static_in.copy_(xs[rank])
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(holder[0].cpu(), torch.stack([x.float() for x in xs]).sum(0).to(torch.bfloat16)), i

        # ---- the fused decode layer at TP = 2 with the group's communicator vs the operator loop over the hooked group ----
        worst = _fused_decode_tp2(rank, world, grp, comm, dev)
        torch.cuda.synchronize()
        assert not comm.timed_out(), "a flag wait gave up"
        dist.barrier()
        tp_hooks.close_all()
        dist.destroy_process_group()
        q.put((rank, "ok", worst))
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


def _fused_decode_tp2(rank, world, grp, comm, dev):
    """One TP = 2 decode step of a 2-layer Llama-3-8B-shaped model: fused_decode.decode_model(model, ..., comm) against the
    layer loop that runs the operators one by one and all-reduces through the HOOKED group (grp.all_reduce)."""
    import dataclasses

    from oracle.layer_parity import ulp_stats
    from sglang_amd import fused_decode, kernels as K
    from sglang_amd.distributed import parallel_state as ps
    from sglang_amd.harness import models as M
    from sglang_amd.harness.engine import Engine, ModelRunner, Req

    # this package's own TP state: rank `rank` of `world`, collectives = the hooked group (what the reference's
    # RowParallelLinear does through tensor_model_parallel_all_reduce -> get_tp_group().all_reduce)
    ps._TP_SIZE, ps._TP_RANK, ps._XGMI = world, rank, comm
    ps._TP_GROUP, ps._TP_CPU_GROUP = grp.device_group, grp.cpu_group
    cfg = dataclasses.replace(M.CONFIGS["llama-3-8b"], num_hidden_layers=2, name="llama-3-8b-2l-tp2")
    B = 16
    runner = ModelRunner(cfg, max_total_tokens=B * 64 + 512, max_running_requests=B, max_context_len=96, device=dev, use_graph=False)
    model = runner.model
    seen = {}
    orig_forward_hidden = model.forward_hidden

    def forward_hidden(input_ids, positions, forward_batch):
        if not forward_batch.forward_mode.is_decode():
            return orig_forward_hidden(input_ids, positions, forward_batch)
        h0 = torch.nn.functional.embedding(input_ids, model.embed_tokens)
        # (a) operator by operator, every row-parallel projection summed by the HOOKED group's all_reduce (what the
        # reference's RowParallelLinear reaches through tensor_model_parallel_all_reduce -> get_tp_group().all_reduce)
        M.OPERATOR_SURFACE_ONLY = True
        plain_ar = ps.tensor_model_parallel_all_reduce
        ps.tensor_model_parallel_all_reduce = lambda t: grp.all_reduce(t)
        try:
            hidden, residual = h0.clone(), None
            for layer in model.layers:
                hidden, residual = layer(positions, hidden, forward_batch, residual)
            want, _ = model.norm(hidden, residual)
        finally:
            ps.tensor_model_parallel_all_reduce = plain_ar
            M.OPERATOR_SURFACE_ONLY = False
        # (b) the plug-in's fused layer loop with the group's communicator
        assert fused_decode.model_fusable(model, h0, forward_batch, comm)
        got = K.unblock(fused_decode.decode_model(model, h0.clone(), positions, forward_batch, comm))
        seen["st"] = ulp_stats(got, want)
        return got

    model.forward_hidden = forward_hidden
    eng = Engine(runner)
    g = torch.Generator().manual_seed(5)
    reqs = [Req(i, torch.randint(0, cfg.vocab_size, (20 + i % 7,), generator=g).tolist(), 3) for i in range(B)]
    eng.prefill(reqs)
    eng.decode_step()
    eng.flush_decode_outputs()
    st = seen["st"]
    assert st["frac_identical"] >= 0.95 and st["frac_within_1ulp"] >= 0.99 and st["max_ulp"] <= 8.0, st
    eng.finish(list(eng.running))
    return st["max_ulp"]


def test_hooked_group_coordinator_two_processes_one_gpu(device):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        for _ in range(world):
            rank, status, payload = q.get(timeout=420)
            assert status == "ok", f"rank {rank}:\n{payload}"
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()


def test_launches_of_one_communicator_are_ordered_across_streams(device):
    """a communicator has one data area and one set of flag counters, so two of its launches must
    never overlap.  Since round 6 every GroupCoordinator has its OWN communicator and the ordering lives in the communicator itself
    (`XgmiAllReduce._ordered`, around every launch whoever the caller is -- hooks, the fused decode layer): a launch that arrives on
    another stream than the previous one waits for an event recorded behind that one.  Observed here with a stand-in communicator:
    stream B's "launch" reads what stream A's "launch" writes behind a long sleep -- it sees the value only if it waited."""
    import types

    import torch

    from sglang_amd.distributed.xgmi_all_reduce import _LaunchOrder

    comm = types.SimpleNamespace(_last_stream=None, _last_event=None, _last_captured=False)
    x = torch.zeros(1, device=device)
    a, b = torch.cuda.Stream(device), torch.cuda.Stream(device)
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        with _LaunchOrder(comm):
            torch.cuda._sleep(200_000_000)            # ~0.1 s of spinning ahead of the write
            x.fill_(1.0)
    with torch.cuda.stream(b):
        with _LaunchOrder(comm):
            y = x.clone()
    torch.cuda.synchronize()
    assert float(y) == 1.0, "stream B's launch did not wait for stream A's"
    assert comm._last_stream == b
    # same stream again: no wait is needed, the event just moves on
    with torch.cuda.stream(b):
        with _LaunchOrder(comm):
            x.add_(1.0)
    torch.cuda.synchronize()
    assert float(x) == 2.0


def test_launches_of_one_communicator_are_ordered_inside_a_capture(device):
    """The same inside ONE stream capture (the reference forks alt streams inside captured forwards for some models): a launch
    on a forked capturing stream gets a graph edge from the previous launch's stream; an eager launch afterwards does not wait
    on anything that was only captured."""
    import types

    import torch

    from sglang_amd.distributed.xgmi_all_reduce import _LaunchOrder

    comm = types.SimpleNamespace(_last_stream=None, _last_event=None, _last_captured=False)
    x = torch.zeros(1, device=device)
    y = torch.zeros(1, device=device)
    side = torch.cuda.Stream(device)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        with _LaunchOrder(comm):
            torch.cuda._sleep(100_000_000)
            x.fill_(3.0)
        side.wait_stream(main)                        # fork
        with torch.cuda.stream(side):
            with _LaunchOrder(comm):                  # previous launch was on `main`: an event edge, not a race
                y.copy_(x)
        main.wait_stream(side)                        # join
    x.zero_(); y.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert float(y) == 3.0
    assert comm._last_captured is True
    with _LaunchOrder(comm):                          # eager, after a capture: nothing to wait for, must not raise
        x.add_(1.0)
    torch.cuda.synchronize()
    assert float(x) == 4.0 and comm._last_captured is False
