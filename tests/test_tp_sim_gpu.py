"""Tensor-parallel forward on ONE GPU: the two TP ranks of a model run as two threads of this process on
their own HIP streams, with tensor_model_parallel_all_reduce / all_gather replaced by an in-process
exchange.  This drives the per-rank GPU code of N > 1 (sharded heads, replicated KV heads, vocab-parallel
head, the unfused projection -> collective -> norm path, 65..128-row GEMM policy) against the TP=1 oracle;
the RCCL transport itself is covered by the driver's multi-GPU run and tests/test_tp_gloo.py (gloo, CPU)."""
import random
import threading

import pytest
import torch

from oracle.model import OracleLM, weights_from_product_model

pytestmark = pytest.mark.gpu


class _FakeGroup:
    """All-reduce / all-gather between `world` threads (the barrier orders the exchange, every thread
    synchronises its own stream before publishing and computes the result on its own stream)."""

    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.local = threading.local()

    def rank(self):
        return self.local.rank

    def exchange(self, x):
        torch.cuda.current_stream().synchronize()
        self.slots[self.rank()] = x
        self.bar.wait()
        parts = list(self.slots)
        self.bar.wait()
        return parts

    def all_reduce(self, x):
        parts = self.exchange(x.clone())
        acc = parts[0].float()
        for p in parts[1:]:
            acc = acc + p.float()           # fp32 sum of the bf16 partials, rounded once (RCCL sums pairwise in bf16)
        x.copy_(acc.to(x.dtype))
        return x

    def all_gather(self, x, dim=-1):
        return torch.cat(self.exchange(x.contiguous()), dim=dim)


def _run_rank(rank, grp, name, prompts, new_tokens, device, out, errs):
    try:
        grp.local.rank = rank
        torch.cuda.set_device(device)
        from sglang_amd.harness.engine import Engine, ModelRunner, Req
        from sglang_amd.harness.models import CONFIGS

        with torch.cuda.stream(torch.cuda.Stream(device=device)):
            runner = ModelRunner(CONFIGS[name], max_total_tokens=4096, max_running_requests=len(prompts) + 2,
                                 max_context_len=256, device=device, init_device="cpu", use_graph=False)
            eng = Engine(runner)
            eng.logits_trace = []
            reqs = [Req(i, p, new_tokens) for i, p in enumerate(prompts)]
            eng.prefill(reqs[:2])                     # two leaders, then the rest hit the radix cache
            eng.prefill(reqs[2:])
            for _ in range(new_tokens - 1):
                eng.decode_step()
            done = sorted(eng.running, key=lambda q: q.rid)
            eng.finish(list(eng.running))
            torch.cuda.current_stream().synchronize()
            out[rank] = (eng.logits_trace, [q.output_ids for q in done], [q.cached_tokens for q in done], runner)
    except Exception as e:  # pragma: no cover
        import traceback

        errs.append(f"rank {rank}: {e}\n{traceback.format_exc()}")
        grp.bar.abort()


@pytest.mark.parametrize("name,world", [("tiny-llama", 2), ("tiny-qwen", 2), ("tiny-llama", 4)])
def test_tp_gpu_forward_matches_tp1_oracle(device, name, world, monkeypatch):
    """world 4 on tiny-llama (2 KV heads): every KV head is replicated on two ranks (llama.py:160-171)."""
    from sglang_amd.distributed import parallel_state as ps
    from sglang_amd.harness.models import CONFIGS, CausalLM

    cfg = CONFIGS[name]
    grp = _FakeGroup(world)
    monkeypatch.setattr(ps, "get_tensor_model_parallel_world_size", lambda: world)
    monkeypatch.setattr(ps, "get_tensor_model_parallel_rank", grp.rank)
    monkeypatch.setattr(ps, "tensor_model_parallel_all_reduce", grp.all_reduce)
    monkeypatch.setattr(ps, "tensor_model_parallel_all_gather", grp.all_gather)
    rnd = random.Random(5)
    shared = [[rnd.randrange(cfg.vocab_size) for _ in range(40)] for _ in range(2)]
    prompts = [shared[i % 2] + [rnd.randrange(cfg.vocab_size) for _ in range(6)] for i in range(6)]
    new_tokens = 4
    out, errs = [None] * world, []
    threads = [threading.Thread(target=_run_rank, args=(r, grp, name, prompts, new_tokens, device, out, errs))
               for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs, errs[0]
    tr0, outs0, hits0, _ = out[0]
    # every rank took identical scheduling decisions and sampled identical tokens
    assert hits0 == [0, 0, 40, 40, 40, 40]
    for tr_r, outs_r, hits_r, _ in out[1:]:
        assert outs_r == outs0 and hits_r == hits0
        for a, b in zip(tr0, tr_r):
            assert torch.equal(a, b)
    # against the unsharded oracle (teacher-forced with the tokens the TP run produced)
    full = CausalLM(cfg, torch.device("cpu"), "cpu", tp_rank=0, tp_size=1)
    oracle = OracleLM(cfg, weights_from_product_model(full), compute_dtype=torch.float32)
    _, ref_logits = oracle.generate(prompts, new_tokens, return_logits=True, forced=outs0)
    got = [torch.cat([tr0[0], tr0[1]])] + tr0[2:]
    for step, (g, r) in enumerate(zip(got, ref_logits)):
        torch.testing.assert_close(g, r, atol=3e-2, rtol=3e-2, msg=f"logits step {step}")
