"""Restatement of the reference decoder forward (torch-native everywhere), used for
end-to-end parity and as bench.py's `cpu_baseline`.  TEST INFRASTRUCTURE ONLY.

`device="cpu"` (default) is the reference's CPU torch-native path (BASELINE configs[0]);
with a HIP device the same plain torch ops run on the GPU -- what the reference's
`--attention-backend torch_native` executes there -- so that the full-size jobs can be
teacher-force-compared in seconds (oracle/parity.py).

Follows /root/reference/python/sglang/srt/models/llama.py:219-223 (rope then
attention), :341-370 (layer wiring with fused add+norm), :419-470 (model loop),
qwen2.py (qkv bias / tied head), mixtral.py:108-118 (router -> topk -> experts),
srt/layers/logits_processor.py:652-700 (last-token logits), with the torch-native
attention backend's write-then-read KV semantics (torch_native_backend.py:279-398)
over a private KV pool and the reference's page_size=1 token allocator order.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import ops


class OracleLM:
    def __init__(self, cfg, weights: Dict[str, torch.Tensor], num_slots: int = 4096, max_ctx: int = 2048,
                 compute_dtype: Optional[torch.dtype] = None, tp_size: int = 1, tp_group=None, max_reqs: int = 63,
                 device="cpu", batched_decode: bool = False, kv_cache_dtype: str = "auto"):
        """`weights`: CPU tensors keyed like the product model's state_dict().  With tp_size > 1 the
        weights are ONE rank's shard (heads / intermediate / vocab split as in the reference's
        Column/RowParallelLinear, srt/layers/linear.py) and the row-parallel outputs are summed with
        torch.distributed.all_reduce over `tp_group` (communication_op.py:18), logits all-gathered
        (logits_processor.py:676)."""
        self.cfg = cfg
        self.tp_size, self.tp_group = tp_size, tp_group
        self.Hq = cfg.num_attention_heads // tp_size
        self.Hkv = max(1, cfg.num_key_value_heads // tp_size)
        self.device = torch.device(device)
        self.w = {k: v.detach().to(self.device) for k, v in weights.items()}     # no copy when already there
        self.compute_dtype = compute_dtype
        self.batched_decode = batched_decode
        D = cfg.head_dim
        if cfg.rope_scaling and cfg.rope_scaling.get("rope_type") == "llama3":
            rs = cfg.rope_scaling
            inv = ops.llama3_inv_freq(D, cfg.rope_theta, rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"],
                                      rs["original_max_position_embeddings"])
        else:
            inv = ops.rope_inv_freq(D, cfg.rope_theta)
        self.rope_cache = ops.cos_sin_cache(inv, cfg.max_position_embeddings).to(torch.bfloat16).to(self.device)
        L = cfg.num_hidden_layers
        dev = self.device
        # "fp8_e4m3": rows are cast to float8_e4m3fn on store (memory_pool.py:2364-2374, scales 1.0 as with dummy weights)
        self.kv_fp8 = kv_cache_dtype == "fp8_e4m3"
        kvd = torch.float8_e4m3fn if self.kv_fp8 else torch.bfloat16
        self.k_cache = [torch.zeros((num_slots, self.Hkv, D), dtype=kvd, device=dev) for _ in range(L)]
        self.v_cache = [torch.zeros((num_slots, self.Hkv, D), dtype=kvd, device=dev) for _ in range(L)]
        self.attn_opts = dict(sliding_window=cfg.sliding_window if getattr(cfg, "sliding_window", None) is not None else -1,
                              logit_cap=float(getattr(cfg, "logit_cap", 0.0) or 0.0))
        self.req_to_token = torch.zeros((max_reqs + 1, max_ctx), dtype=torch.int32, device=dev)
        self.next_slot = 1
        # tests/test_layer_parity_gpu.py: set to [] to record, per forward() call, every layer's inputs and outputs
        # (the tensors a product layer is then fed with: identical-input, per-layer comparison)
        self.trace: Optional[list] = None
        # a trace recorded by ANOTHER OracleLM run of the same job: when set, every layer of call k starts from that
        # run's layer inputs (h_in, res_in) instead of its own -- per-layer comparison of two evaluations of the graph
        self.inject: Optional[list] = None
        self._calls = 0
        # {layer: int tensor [requests, max_len, top_k]}: expert ids to use instead of the oracle's own top-k choice for
        # the token of request b (req_pool index b + 1) at position t -- a whole-depth MoE comparison with the routing
        # of the run under test, so that near-tie flips of the discrete choice do not decorrelate the two runs
        self.forced_topk_ids: Optional[Dict[int, torch.Tensor]] = None

    # ---- one forward over a ragged batch --------------------------------------------------
    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, req_pool: torch.Tensor, seq_lens: torch.Tensor,
                prefix_lens: torch.Tensor, extend_lens: torch.Tensor, out_loc: torch.Tensor, decode: bool,
                input_embeds: Optional[torch.Tensor] = None) -> torch.Tensor:
        cfg, w = self.cfg, self.w
        D, Hq, Hkv = cfg.head_dim, self.Hq, self.Hkv
        h = input_embeds if input_embeds is not None else F.embedding(input_ids, w["embed_tokens"])
        residual = None
        rec = None
        if self.trace is not None:
            rec = dict(decode=decode, positions=positions, out_loc=out_loc, layers=[])
            self.trace.append(rec)
        inj = self.inject[self._calls]["layers"] if self.inject is not None else None
        self._calls += 1
        for i in range(cfg.num_hidden_layers):
            p = f"layers.{i}."
            if inj is not None:
                h, residual = inj[i]["h_in"], inj[i]["res_in"]
            lr = None
            if rec is not None:
                lr = dict(h_in=h, res_in=residual)
                rec["layers"].append(lr)
            if residual is None:
                residual = h
                h = ops.rmsnorm(h, w[p + "input_layernorm.weight"], cfg.rms_norm_eps)
            else:
                h, residual = ops.fused_add_rmsnorm(h, residual, w[p + "input_layernorm.weight"], cfg.rms_norm_eps)
            if lr is not None:
                lr.update(normed=h, residual=residual)
            qkv = F.linear(h, w[p + "self_attn.qkv_proj.weight"], w.get(p + "self_attn.qkv_proj.bias"))
            q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
            q, k = ops.rotary_embedding(positions, q, k, D, self.rope_cache, True)
            if lr is not None:
                lr.update(qkv=qkv, q_rot=q, k_rot=k, v=v)
            if self.kv_fp8:
                ops.store_kv(ops.quantize_kv_fp8(k.reshape(-1, Hkv, D)), ops.quantize_kv_fp8(v.reshape(-1, Hkv, D)),
                             self.k_cache[i], self.v_cache[i], out_loc)
            else:
                ops.store_kv(k.reshape(-1, Hkv, D), v.reshape(-1, Hkv, D), self.k_cache[i], self.v_cache[i], out_loc)
            q3 = q.reshape(-1, Hq, D)
            if decode:
                o = ops.decode_attention(q3, self.k_cache[i], self.v_cache[i], self.req_to_token, req_pool, seq_lens,
                                         D ** -0.5, self.compute_dtype, batched=self.batched_decode, **self.attn_opts)
            else:
                o = ops.extend_attention(q3, self.k_cache[i], self.v_cache[i], self.req_to_token, req_pool, seq_lens,
                                         prefix_lens, extend_lens, D ** -0.5, True, self.compute_dtype, **self.attn_opts)
            h = self._all_reduce(F.linear(o.reshape(-1, Hq * D), w[p + "self_attn.o_proj.weight"]))
            if lr is not None:
                lr.update(o_proj=h)
            h, residual = ops.fused_add_rmsnorm(h, residual, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
            if lr is not None:
                lr.update(post_normed=h)
            if cfg.num_local_experts > 0:
                logits = F.linear(h, w[p + "mlp.gate.weight"])
                tw, ti = ops.fused_topk(logits, cfg.num_experts_per_tok, True)
                if self.forced_topk_ids is not None:
                    rows = req_pool if decode else torch.repeat_interleave(req_pool, extend_lens)
                    ti = self.forced_topk_ids[i][rows - 1, positions].to(ti.dtype)
                    tw = ops.topk_weights_for_ids(logits, ti, True)
                if lr is not None:
                    lr.update(topk_ids=ti, topk_weights=tw, router_logits=logits)
                h = self._all_reduce(ops.moe_forward(h, w[p + "mlp.experts.w13_weight"], w[p + "mlp.experts.w2_weight"], tw, ti))
            else:
                gu = F.linear(h, w[p + "mlp.gate_up_proj.weight"])
                act = ops.silu_and_mul(gu)
                if lr is not None:
                    lr.update(act=act)
                h = self._all_reduce(F.linear(act, w[p + "mlp.down_proj.weight"]))
            if lr is not None:
                lr.update(attn_out=o, out=h, res_out=residual)
        h, final_res = ops.fused_add_rmsnorm(h, residual, w["norm.weight"], cfg.rms_norm_eps)
        if rec is not None:
            rec.update(final_normed=h, final_residual=final_res)
        if not decode:
            last = torch.cumsum(extend_lens, 0) - 1
            h = h[last]
        logits = F.linear(h, w["lm_head"])
        if self.tp_size > 1:
            import torch.distributed as dist

            parts = [torch.empty_like(logits) for _ in range(self.tp_size)]
            dist.all_gather(parts, logits.contiguous(), group=self.tp_group)
            logits = torch.cat(parts, dim=-1)
        if rec is not None:
            rec["logits"] = logits
        return logits.float()

    def _all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        if self.tp_size > 1:
            import torch.distributed as dist

            x = x.contiguous()
            dist.all_reduce(x, group=self.tp_group)
        return x

    # ---- greedy generation with prefix reuse expressed the plain way ------------------------
    def generate(self, prompts: Sequence[Sequence[int]], max_new_tokens: int, return_logits: bool = False,
                 forced: Optional[Sequence[Sequence[int]]] = None, logits_hook=None,
                 share_prefix_groups: Optional[Sequence[Sequence[int]]] = None, shared_len: int = 0,
                 prompt_embeds: Optional[Sequence[Optional[torch.Tensor]]] = None):
        """Every request gets its own fresh slots (no sharing): the reference result the
        radix-cached run must reproduce.  `forced[b][i]` (teacher forcing) replaces the
        oracle's own i-th sampled token as the next input, so logits can be compared
        step by step against a run whose argmax flipped on a near-tie.  `logits_hook(step, logits)`
        receives every step's fp32 logits instead of keeping them all (full-size jobs).

        `share_prefix_groups` (lists of prompt indices whose first `shared_len` tokens are equal) runs the
        prefill the way the radix-cached scheduler does -- each group's first request cold, the others as
        extends over that request's slots for the shared part -- the phases bench.py times (cpu_baseline)."""
        dev = self.device
        B = len(prompts)
        req_pool = torch.arange(1, B + 1, device=dev)
        lens_l = [len(p) for p in prompts]
        lens = torch.tensor(lens_l, device=dev)
        logits = torch.empty((B, self.w["lm_head"].shape[0] * self.tp_size), dtype=torch.float32, device=dev)
        if share_prefix_groups:
            passes = [[g[0] for g in share_prefix_groups], [b for g in share_prefix_groups for b in g[1:]]]
            leader_of = {b: g[0] for g in share_prefix_groups for b in g[1:]}
        else:
            passes, leader_of = [list(range(B))], {}
        for members in passes:
            if not members:
                continue
            pre_l = [shared_len if b in leader_of else 0 for b in members]
            ext_l = [lens_l[b] - p for b, p in zip(members, pre_l)]
            T = sum(ext_l)
            out_loc = torch.arange(self.next_slot, self.next_slot + T, device=dev)
            self.next_slot += T
            off = 0
            for b, pre, ext in zip(members, pre_l, ext_l):
                if pre:
                    self.req_to_token[b + 1, :pre] = self.req_to_token[leader_of[b] + 1, :pre]
                self.req_to_token[b + 1, pre: pre + ext] = out_loc[off: off + ext].to(torch.int32)
                off += ext
            ids = torch.tensor([t for b, pre in zip(members, pre_l) for t in prompts[b][pre:]], device=dev)
            emb = None
            if prompt_embeds is not None:
                # image + text prompts: the caller supplies the whole prompt's embeddings (oracle/vision.py
                # embed_with_images); ids of image positions are pad values outside the vocabulary
                vocab = self.w["embed_tokens"].shape[0]
                emb = torch.cat([(prompt_embeds[b][pre:].to(dev) if prompt_embeds[b] is not None
                                  else F.embedding(torch.tensor(prompts[b][pre:], device=dev).clamp(0, vocab - 1), self.w["embed_tokens"]))
                                 for b, pre in zip(members, pre_l)]).to(self.w["embed_tokens"].dtype)
            pos = torch.cat([torch.arange(pre, pre + ext, device=dev) for pre, ext in zip(pre_l, ext_l)])
            midx = torch.tensor(members, device=dev)
            logits[midx] = self.forward(ids, pos, req_pool[midx], lens[midx], torch.tensor(pre_l, device=dev),
                                        torch.tensor(ext_l, device=dev), out_loc, decode=False, input_embeds=emb)
        all_logits = []
        if logits_hook is not None:
            logits_hook(0, logits)
        if return_logits:
            all_logits.append(logits)
        first = logits.argmax(-1).tolist()
        outs = [[int(t)] for t in first]
        fed = [[forced[b][0]] if forced is not None else [outs[b][0]] for b in range(B)]
        seq = lens.clone()
        for step in range(1, max_new_tokens):
            loc = torch.arange(self.next_slot, self.next_slot + B, device=dev)
            self.next_slot += B
            self.req_to_token[req_pool, seq] = loc.to(torch.int32)
            seq = seq + 1
            last = torch.tensor([f[-1] for f in fed], device=dev)
            logits = self.forward(last, seq - 1, req_pool, seq, None, None, loc, decode=True)
            if logits_hook is not None:
                logits_hook(step, logits)
            if return_logits:
                all_logits.append(logits)
            for b, (o, t) in enumerate(zip(outs, logits.argmax(-1).tolist())):
                o.append(int(t))
                fed[b].append(forced[b][len(fed[b])] if forced is not None else int(t))
        return (outs, all_logits) if return_logits else outs


def weights_from_product_model(model, device="cpu") -> Dict[str, torch.Tensor]:
    """Collect the product model's (TP=1) parameters under oracle names (on `device`; no copy when the
    parameters already live there)."""
    w = {"embed_tokens": model.embed_tokens.data, "norm.weight": model.norm.weight.data, "lm_head": model.lm_head.data}
    for i, layer in enumerate(model.layers):
        p = f"layers.{i}."
        w[p + "input_layernorm.weight"] = layer.input_layernorm.weight.data
        w[p + "post_attention_layernorm.weight"] = layer.post_attention_layernorm.weight.data
        w[p + "self_attn.qkv_proj.weight"] = layer.self_attn.qkv_proj.weight.data
        if layer.self_attn.qkv_proj.bias is not None:
            w[p + "self_attn.qkv_proj.bias"] = layer.self_attn.qkv_proj.bias.data
        w[p + "self_attn.o_proj.weight"] = layer.self_attn.o_proj.weight.data
        mlp = layer.mlp
        if hasattr(mlp, "gate_up_proj"):
            w[p + "mlp.gate_up_proj.weight"] = mlp.gate_up_proj.weight.data
            w[p + "mlp.down_proj.weight"] = mlp.down_proj.weight.data
        else:
            w[p + "mlp.gate.weight"] = mlp.gate.weight.data
            w[p + "mlp.experts.w13_weight"] = mlp.experts.w13_weight.data
            w[p + "mlp.experts.w2_weight"] = mlp.experts.w2_weight.data
    return {k: v.detach().to(device) for k, v in w.items()}
