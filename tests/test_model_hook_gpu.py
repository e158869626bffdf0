"""The fused decode step through the hook plugin.load() installs on the reference's LlamaModel.forward / Qwen2Model.forward
(sglang_amd/fused_decode.py; srt/plugins/hook_registry.py AROUND semantics: hook(original_fn, self, *args)).

A model object with the REFERENCE's attribute layout and forward loop (llama.py:419-470 restated below, driving the
layers operator by operator) is evaluated twice on the same live decode batch of the engine: through the hook (which
must take the fused 9-launch branch and never call the original) and through the original loop.  Same rounding points
on both paths -> the hidden states agree to one bf16 ulp (of the value, floored at the row's rms), >= 97 % bit for bit."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


class RefShapedLlamaModel:
    """What the hook sees under sglang: LlamaModel's attributes, its forward signature and its layer loop."""

    def __init__(self, causal_lm):
        self.layers, self.norm = causal_lm.layers, causal_lm.norm
        self._embed = causal_lm.embed_tokens
        self.pp_group = types.SimpleNamespace(is_first_rank=True, is_last_rank=True)
        self.start_layer, self.end_layer, self.layers_to_capture = 0, len(self.layers), []

    def embed_tokens(self, input_ids):
        return torch.nn.functional.embedding(input_ids, self._embed)

    def forward(self, input_ids, positions, forward_batch, input_embeds=None, pp_proxy_tensors=None):   # llama.py:419-470
        hidden_states = self.embed_tokens(input_ids) if input_embeds is None else input_embeds
        residual = None
        for i in range(self.start_layer, self.end_layer):
            hidden_states, residual = self.layers[i](positions, hidden_states, forward_batch, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


# (the third case: the 8B shapes with a qkv bias -- the layer form of Qwen2Model, the hook's second target)
# (the last two: sparse-MoE layers -- the form of MixtralModel, the hook's third target: fused attention half, the block's own gate /
# TopK / experts, add + norm in one launch)
# (("llama-3-8b-2l", 128 / 80): the hybrid form -- from 65 rows the wide gate_up projection is the library's GEMM + silu_and_mul, the rest
# of the layer keeps its weight streams and fused combines)
@pytest.mark.parametrize("model_name,B", [("tiny-llama3-rope", 5), ("llama-3-8b-2l", 64), ("llama-3-8b-2l-qkvbias", 16),
                                          ("tiny-mixtral", 5), ("mixtral-8x7b-2l", 32), ("llama-3-8b-2l", 128), ("llama-3-8b-2l", 80)])
def test_hook_takes_the_fused_branch_and_matches_the_operator_loop(device, monkeypatch, model_name, B):
    import dataclasses

    from sglang_amd import fused_decode
    from sglang_amd.harness import models as M
    from sglang_amd.harness.engine import Engine, ModelRunner, Req

    cfg = (dataclasses.replace(M.CONFIGS["llama-3-8b"], num_hidden_layers=2, name=model_name, attention_bias=model_name.endswith("qkvbias"))
           if model_name.startswith("llama-3-8b-2l") else
           dataclasses.replace(M.CONFIGS["mixtral-8x7b"], num_hidden_layers=2, name=model_name) if model_name == "mixtral-8x7b-2l" else M.CONFIGS[model_name])
    runner = ModelRunner(cfg, max_total_tokens=B * 80 + 512, max_running_requests=B, max_context_len=96, device=device, use_graph=False)
    ref_model = RefShapedLlamaModel(runner.model)
    seen = {}

    def forward_hidden(input_ids, positions, forward_batch):
        if not forward_batch.forward_mode.is_decode():
            return ref_model.forward(input_ids, positions, forward_batch)
        # the operator-by-operator loop (what the reference runs without the hook) ...
        monkeypatch.setattr(M, "OPERATOR_SURFACE_ONLY", True)
        want = ref_model.forward(input_ids, positions, forward_batch)
        monkeypatch.setattr(M, "OPERATOR_SURFACE_ONLY", False)
        # ... and the hooked forward: AROUND = hook(original_fn, self, *args)
        called = []

        def original(self, *a, **k):
            called.append(1)
            return RefShapedLlamaModel.forward(self, *a, **k)

        got = fused_decode.llama_model_forward_hook(original, ref_model, input_ids, positions, forward_batch)
        seen["fused"] = not called
        seen["want"], seen["got"] = want.clone(), got.clone()
        return got

    monkeypatch.setattr(runner.model, "forward_hidden", forward_hidden)
    eng = Engine(runner)
    g = torch.Generator().manual_seed(3)
    reqs = [Req(i, torch.randint(0, cfg.vocab_size, (20 + i % 7,), generator=g).tolist(), 3) for i in range(B)]
    eng.prefill(reqs)
    eng.decode_step()
    eng.flush_decode_outputs()
    assert seen.get("fused") is True, "the hook fell back to the original forward on a batch it owns"
    from oracle.layer_parity import ulp_stats          # bf16 ulps of max(|ref|, row rms)

    st = ulp_stats(seen["got"], seen["want"])
    # (the two paths pick different split-K decompositions for o_proj / down_proj -- with and without the norm in the
    # combine -- so a few sums round the other way and two layers carry that on: measured 98.1 % identical, 99.66 %
    # within one ulp, 99.94 % within two at the 8B shapes)
    if cfg.num_local_experts > 0:
        # a one-ulp difference in a router input may move a token to another expert in either evaluation (a whole row then differs by
        # far more than rounding): judged over the rows whose routing agreed -- all but at most one row of these small batches
        rows_ok = ((seen["got"].float() - seen["want"].float()).abs().amax(-1) <= 0.05 * seen["want"].float().abs().amax(-1).clamp_min(1e-3))
        assert int(rows_ok.sum()) >= B - 1, rows_ok
        st = ulp_stats(seen["got"][rows_ok], seen["want"][rows_ok])
        assert st["frac_identical"] >= 0.90 and st["frac_within_2ulp"] >= 0.99 and st["max_ulp"] <= 16.0, st
    else:
        assert st["frac_identical"] >= 0.95 and st["frac_within_1ulp"] >= 0.99 and st["frac_within_2ulp"] >= 0.998 and st["max_ulp"] <= 8.0, st
    eng.finish(list(eng.running))


def test_linear_hook_streams_decode_sized_projections_and_leaves_the_rest(device):
    """The AROUND hook on UnquantizedLinearMethod.apply (sglang_amd/linear_hook.py): decode-sized bf16 projections -- with
    and without a bias, 2-D and 3-D inputs -- run the weight-streaming GEMM (the reference's method is not entered) and
    equal F.linear to one bf16 ulp; a prefill-sized batch and an fp32 input reach the reference's method."""
    from sglang_amd import linear_hook

    g = torch.Generator().manual_seed(21)
    called = []

    def original(self, layer, x, bias=None):
        called.append(x.shape)
        return torch.nn.functional.linear(x, layer.weight, bias)

    for N, K in ((6144, 4096), (4096, 1024), (1536, 896)):
        w = torch.nn.Parameter((torch.randn((N, K), generator=g) * 0.02).to(BF).to(device), requires_grad=False)
        layer = types.SimpleNamespace(weight=w)
        b = (torch.randn(N, generator=g) * 0.1).to(BF).to(device)
        for shape, bias in (((64, K), None), ((5, K), b), ((2, 7, K), None)):
            x = torch.randn(shape, generator=g).to(BF).to(device)
            n = len(called)
            got = linear_hook.unquant_apply_hook(original, None, layer, x, bias)
            assert len(called) == n, "the reference's method was entered for a decode-sized projection"
            # (fp64 on the host: an fp32 GEMM on the device loads another library for one comparison)
            want = torch.nn.functional.linear(x.double().cpu(), w.double().cpu(), bias.double().cpu() if bias is not None else None)
            assert got.shape == want.shape
            err = (got.double().cpu() - want).abs()
            assert float((err > 2.0 ** -8 * want.abs() + 1e-3).float().mean()) == 0.0, float(err.max())
        big = torch.randn((512, K), generator=g).to(BF).to(device)
        n = len(called)
        linear_hook.unquant_apply_hook(original, None, layer, big)
        assert len(called) == n + 1                                       # prefill-sized: the library GEMM
    x32 = torch.randn((4, 896), generator=g).to(device)
    w32 = types.SimpleNamespace(weight=torch.nn.Parameter(torch.randn((1536, 896), generator=g).to(device), requires_grad=False))
    n = len(called)
    linear_hook.unquant_apply_hook(original, None, w32, x32)
    assert len(called) == n + 1
