"""The TP=1 decode step of a Llama-style decoder as 9 launches per layer, on ANY model object wired like the reference's.

What `LlamaModel.forward` (/root/reference/python/sglang/srt/models/llama.py:419-470) computes for a decode batch --
per layer `input_layernorm -> qkv_proj -> rotary_emb (+ KV store) -> attn -> o_proj -> post_attention_layernorm ->
gate_up_proj -> act_fn -> down_proj` (:341-370), then `norm` -- with every residual-add + RMSNorm executed by the
PRECEDING projection's split-K combine kernel, rope + the KV-row store by the qkv projection's combine, SiLU-mul in the
gate_up GEMM's epilogue:

    wstream_qkv_rope -> attention (+ merge) -> o_proj GEMM + combine(add, norm) -> gate_up GEMM (silu epilogue)
    -> down_proj GEMM + combine(add, NEXT layer's input norm)

Same arithmetic and rounding points as the operator-by-operator path (tests/test_layer_parity_gpu.py feeds both with
the oracle's inputs).  The functions below read the model through the attribute names the reference's classes carry
(`layers[i].self_attn.{qkv_proj, o_proj, rotary_emb, attn}`, `.mlp.{gate_up_proj, down_proj}`, `.input_layernorm`,
`.post_attention_layernorm`, `norm`, `embed_tokens`), so they run on the reference's `LlamaModel` unchanged --
`plugin.load()` installs `llama_model_forward_hook` on it through the reference's own HookRegistry
(srt/plugins/hook_registry.py:84 register, :146 apply_hooks; HookType.AROUND) -- and on this package's harness
models (harness/models.py), which is how bench.py measures exactly what the plug-in delivers.

Everything outside this form (prefill, TP > 1 without the xGMI communicator, pipeline stages, quantised or biased-MLP layers, MoE, captured aux
hidden states, batches the weight-streaming GEMM does not take) goes to the original forward -- the reference's own
code -- so the hook can never change a result it does not own.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import kernels

_BF16 = torch.bfloat16


# (decoder layer, attention, MLP) classes whose forwards are exactly the form decode_layer restates: llama.py:70-136 /
# :138-281 / :283-370 and qwen2.py:63-124 / :126-235 / :237-313 (this package's harness classes carry the Llama names).
# The test is on the EXACT class: a subclass that inherits the hooked model forward but changes the attention math
# (Ministral3Attention scales q, llama.py users with per-head norms, ...) is not this form and stays with the reference.
FUSABLE_FORMS = (("LlamaDecoderLayer", "LlamaAttention", "LlamaMLP"), ("Qwen2DecoderLayer", "Qwen2Attention", "Qwen2MLP"))
# Sparse-MoE layers (round 5): the ATTENTION half is the same four fused launches (qkv + rope / store, attention, o_proj + add /
# norm); the MoE block behind it runs through its own pieces -- replicated gate, TopK, FusedMoE (the grouped GEMMs in the fused-MoE
# slot) -- and the residual add + next layer's input norm behind it is one launch.  mixtral.py:57-261 (MixtralMoE: `gate`, `topk`,
# `experts`, an all-reduce at TP > 1); this package's harness layer carries the block as `mlp` (harness/moe_block.py).
MOE_FORMS = (("MixtralDecoderLayer", "MixtralAttention", "MixtralMoE"), ("LlamaDecoderLayer", "LlamaAttention", "SparseMoeBlock"))


def moe_block_of(layer):
    """The sparse-MoE block of a decoder layer (mixtral.py:224 `block_sparse_moe`; the harness layer's `mlp`), or None."""
    blk = getattr(layer, "block_sparse_moe", None)
    if blk is None:
        blk = getattr(layer, "mlp", None)
    return blk if blk is not None and all(hasattr(blk, n) for n in ("gate", "topk", "experts")) else None


def _plain_linear(lin) -> bool:
    """An unquantised bf16 projection (UnquantizedLinearMethod, linear.py:1596-1660): weight [N, K] bf16, K contiguous.
    A LoRA wrapper (lora/layers.py:39-50 BaseLayerWithLoRA: `.weight` aliases the base weight, no quant_method) is NOT one:
    streaming its base weights would drop the adapter's delta."""
    if hasattr(lin, "base_layer") or hasattr(lin, "set_lora") or hasattr(lin, "lora_backend"):
        return False
    w = getattr(lin, "weight", None)
    if w is None or w.dtype != _BF16 or w.dim() != 2 or w.stride(1) != 1 or not w.is_cuda:
        return False
    qm = getattr(lin, "quant_method", None)
    return qm is None or type(qm).__name__ == "UnquantizedLinearMethod"


def layer_unfusable_reason(layer, rows: int) -> Optional[str]:
    """Why this layer is NOT a dense Llama-style block whose four projections the weight-streaming GEMM takes at `rows` rows
    (None when it is).  Also what `explain()` reports to someone asking why a model stays on the operator-by-operator path."""
    attn, mlp = getattr(layer, "self_attn", None), getattr(layer, "mlp", None)
    moe = moe_block_of(layer)
    if attn is not None and moe is not None:
        form = (type(layer).__name__, type(attn).__name__, type(moe).__name__)
        if form not in MOE_FORMS:
            return f"layer classes {form} are not one of {MOE_FORMS}"
        lins = dict(qkv_proj=attn.qkv_proj, o_proj=attn.o_proj)
        mlp = None
    else:
        if attn is None or mlp is None or not hasattr(mlp, "gate_up_proj") or not hasattr(mlp, "down_proj"):
            return "no self_attn / mlp.gate_up_proj / mlp.down_proj"
        form = (type(layer).__name__, type(attn).__name__, type(mlp).__name__)
        if form not in FUSABLE_FORMS:
            return f"layer classes {form} are not one of {FUSABLE_FORMS}"
        lins = dict(qkv_proj=attn.qkv_proj, o_proj=attn.o_proj, gate_up_proj=mlp.gate_up_proj, down_proj=mlp.down_proj)
    for name, l in lins.items():
        if not _plain_linear(l):
            w = getattr(l, "weight", None)
            return (f"{name} is not a plain bf16 projection (weight {getattr(w, 'dtype', None)} {tuple(getattr(w, 'shape', ()))} "
                    f"strides {w.stride() if w is not None else None} cuda {getattr(w, 'is_cuda', None)}, quant_method "
                    f"{type(getattr(l, 'quant_method', None)).__name__})")
    if getattr(attn.o_proj, "bias", None) is not None or (mlp is not None and (
            getattr(mlp.gate_up_proj, "bias", None) is not None or getattr(mlp.down_proj, "bias", None) is not None)):
        return "o_proj / gate_up_proj / down_proj carry a bias"
    rope = attn.rotary_emb
    # plain cos / sin-cache ropes only (rotary_embedding/base.py:78 RotaryEmbedding, rope_variant.py:537
    # Llama3RotaryEmbedding: a scaled cache, the same forward); multimodal / long-rope / dual-chunk variants index the
    # cache differently and stay with the reference
    if type(rope).__name__ not in ("RotaryEmbedding", "Llama3RotaryEmbedding"):
        return f"rope class {type(rope).__name__}"
    if not getattr(rope, "is_neox_style", False) or getattr(rope, "rotary_dim", attn.head_dim) != attn.head_dim:
        return "rope is not neox-style over the whole head"
    if any(hasattr(attn, n) for n in ("q_norm", "k_norm")):            # (qwen3-style per-head norms: another layer form)
        return "per-head q / k norms"
    hidden = attn.qkv_proj.weight.shape[1]
    if hidden % 128 != 0 or hidden > 16384 or (mlp is not None and mlp.gate_up_proj.weight.shape[0] % 32 != 0):
        return f"hidden size {hidden} / gate_up rows outside the kernels' tiling"
    for name, l in lins.items():
        if name == "gate_up_proj" and kernels.wstream_supported(rows, *l.weight.shape):
            continue     # at 65..128 rows the WIDE projection is the library's (wstream_preferred): decode_layer's hybrid form
        if not kernels.wstream_preferred(rows, *l.weight.shape):
            return f"{name} {tuple(l.weight.shape)} at {rows} rows is left to the library GEMM"
    return None


def layer_fusable(layer, rows: int) -> bool:
    """The layer is a dense Llama-style block whose four projections the weight-streaming GEMM takes at `rows` rows."""
    return layer_unfusable_reason(layer, rows) is None


def kv_pool_of(forward_batch):
    """The KV pool of the running forward.  The reference does not hang the pools on the ForwardBatch: they are resolved through the
    active attention backend (model_executor/forward_context.py:70-76 get_token_to_kv_pool -> get_attn_backend().token_to_kv_pool,
    published by ModelRunner._forward_raw and by the graph runners around model.forward); this package's own harness batch
    carries the pool itself."""
    pool = getattr(forward_batch, "token_to_kv_pool", None)
    if pool is not None:
        return pool
    try:
        from sglang.srt.model_executor.forward_context import get_token_to_kv_pool, has_forward_context

        if has_forward_context():
            return get_token_to_kv_pool()
    except ImportError:
        pass
    backend = getattr(forward_batch, "attn_backend", None)
    pool = getattr(backend, "token_to_kv_pool", None)
    if pool is None:
        raise AttributeError("no KV pool: neither forward_batch.token_to_kv_pool, an active forward context, nor forward_batch.attn_backend")
    return pool


def decode_layer(layer, positions: torch.Tensor, normed: torch.Tensor, forward_batch, residual: torch.Tensor,
                 next_norm, comm=None) -> torch.Tensor:
    """One layer of the fused form: `normed` = this layer's input_layernorm output (row-major or chunk-major), `residual`
    the residual stream (updated in place); returns next_norm(residual') -- chunk-major at TP = 1 -- for the next layer /
    lm_head.  `comm` (TP > 1): the TP group's xGMI communicator; the row-parallel projections (o_proj, down_proj:
    linear.py RowParallelLinear, this rank's K shard) then produce partial sums and the all-reduce that follows carries
    the residual add + RMSNorm in its epilogue -- the layer keeps its 9 launches."""
    from .layers.attention.hip_backend import pool_kernel_format

    attn, mlp = layer.self_attn, layer.mlp
    pool = kv_pool_of(forward_batch)
    layer_id = attn.attn.layer_id
    fmt = pool_kernel_format(pool, attn.attn)
    plain = not fmt["kv_fp8"] and not fmt["hnd"]
    bias = getattr(attn.qkv_proj, "bias", None)
    q = kernels.wstream_qkv_rope(normed, attn.qkv_proj.weight.data, bias.data if bias is not None else None, positions,
                                 attn.rotary_emb.cos_sin_cache, attn.num_heads, attn.num_kv_heads, attn.head_dim,
                                 pool.get_key_buffer(layer_id), pool.get_value_buffer(layer_id), forward_batch.out_cache_loc,
                                 **({} if plain else fmt))
    a = attn.attn(q, None, None, forward_batch, save_kv_cache=False)
    post = layer.post_attention_layernorm
    blocked_act = mlp.gate_up_proj.weight.shape[0] % 256 == 0
    w_gu = mlp.gate_up_proj.weight.data
    # 65..128 rows: the weight stream still wins for qkv / o / down, the wide gate_up is the library's (kernels.wstream_preferred):
    # the layer keeps its fused combines around a library GEMM + silu_and_mul -- 11 launches instead of the operator path's 15
    stream_gu = kernels.wstream_preferred(a.shape[0], *w_gu.shape)

    def gate_up(x):
        if stream_gu:
            return kernels.wstream_gemm(x, w_gu, epilogue="silu_and_mul", out_blocked=blocked_act)
        return kernels.silu_and_mul(torch.nn.functional.linear(x, w_gu))

    if comm is not None:
        y = kernels.wstream_gemm(a.reshape(a.shape[0], -1), attn.o_proj.weight.data)
        x = comm.all_reduce_add_rmsnorm(y, residual, post.weight.data, post.variance_epsilon)
        y = kernels.wstream_gemm(gate_up(x), mlp.down_proj.weight.data)
        return comm.all_reduce_add_rmsnorm(y, residual, next_norm.weight.data, next_norm.variance_epsilon)
    x = kernels.wstream_gemm(a, attn.o_proj.weight.data, epilogue="add_rmsnorm", residual=residual, norm_weight=post.weight.data,
                             eps=post.variance_epsilon, out_blocked=stream_gu)
    return kernels.wstream_gemm(gate_up(x), mlp.down_proj.weight.data, epilogue="add_rmsnorm", residual=residual,
                                norm_weight=next_norm.weight.data, eps=next_norm.variance_epsilon, out_blocked=True)


def decode_layer_moe(layer, positions: torch.Tensor, normed: torch.Tensor, forward_batch, residual: torch.Tensor,
                     next_norm, comm=None) -> torch.Tensor:
    """A sparse-MoE decoder layer (mixtral.py:238-260) in the fused form: the attention half as in decode_layer() -- four
    launches -- handing ROW-MAJOR normed activations to the MoE block's own pieces (gate -> TopK -> FusedMoE: the reference's
    modules with the registered forwards / the fused-MoE slot), then the residual add + `next_norm` in one launch (at TP > 1: in
    the epilogue of the all-reduce that follows the experts, mixtral.py:115-117).  Returns next_norm(residual'), row-major."""
    from .layers.attention.hip_backend import pool_kernel_format

    attn, moe = layer.self_attn, moe_block_of(layer)
    pool = kv_pool_of(forward_batch)
    layer_id = attn.attn.layer_id
    fmt = pool_kernel_format(pool, attn.attn)
    plain = not fmt["kv_fp8"] and not fmt["hnd"]
    bias = getattr(attn.qkv_proj, "bias", None)
    q = kernels.wstream_qkv_rope(normed, attn.qkv_proj.weight.data, bias.data if bias is not None else None, positions,
                                 attn.rotary_emb.cos_sin_cache, attn.num_heads, attn.num_kv_heads, attn.head_dim,
                                 pool.get_key_buffer(layer_id), pool.get_value_buffer(layer_id), forward_batch.out_cache_loc,
                                 **({} if plain else fmt))
    a = attn.attn(q, None, None, forward_batch, save_kv_cache=False)
    post = layer.post_attention_layernorm
    if comm is not None:
        y = kernels.wstream_gemm(a.reshape(a.shape[0], -1), attn.o_proj.weight.data)
        x = comm.all_reduce_add_rmsnorm(y, residual, post.weight.data, post.variance_epsilon)
    else:
        x = kernels.wstream_gemm(a, attn.o_proj.weight.data, epilogue="add_rmsnorm", residual=residual, norm_weight=post.weight.data,
                                 eps=post.variance_epsilon)
    r = moe.gate(x)
    router_logits = r[0] if isinstance(r, tuple) else r              # (ReplicatedLinear returns (output, bias))
    h = moe.experts(x, moe.topk(x, router_logits))
    if comm is not None:
        return comm.all_reduce_add_rmsnorm(h.contiguous(), residual, next_norm.weight.data, next_norm.variance_epsilon)
    kernels.fused_add_rmsnorm(h, residual, next_norm.weight.data, next_norm.variance_epsilon)      # h <- norm(h + residual), in place
    return h


def plain_embedding_weight(emb) -> Optional[torch.Tensor]:
    """The table of an embedding module whose forward at TP = 1 is exactly `F.embedding(ids, weight)` (vocab_parallel_embedding.py:
    527-540, 566-579: an unsharded VocabParallelEmbedding with the unquantised method), else None: sharded tables (masking + the
    all-reduce), quantised or LoRA-wrapped embeddings keep the module's own forward."""
    if type(emb).__name__ != "VocabParallelEmbedding" or hasattr(emb, "base_layer") or hasattr(emb, "set_lora"):
        return None
    if int(getattr(emb, "tp_size", 1) or 1) != 1:
        return None
    qm = getattr(emb, "quant_method", None)
    if qm is not None and type(qm).__name__ != "UnquantizedEmbeddingMethod":
        return None
    w = getattr(emb, "weight", None)
    if not isinstance(w, torch.Tensor) or w.dtype != _BF16 or w.dim() != 2 or w.stride(1) != 1 or not w.is_cuda or w.data_ptr() % 16 or w.stride(0) % 8:
        return None
    return w.data


def embed_and_norm(model, input_ids: torch.Tensor):
    """(embed_tokens(input_ids), input_layernorm of layer 0 of it) -- one launch when the embedding is a plain table and the ids are the
    int64 vector a decode batch carries; the module's own forward + the norm kernel otherwise."""
    n0 = model.layers[0].input_layernorm
    w = plain_embedding_weight(model.embed_tokens)
    if (w is not None and isinstance(input_ids, torch.Tensor) and input_ids.is_cuda and input_ids.dtype == torch.int64 and input_ids.dim() == 1
            and input_ids.is_contiguous() and n0.weight.dtype == _BF16 and n0.weight.shape[-1] == w.shape[1]
            and int(getattr(model.embed_tokens, "num_embeddings", w.shape[0])) <= w.shape[0]):
        return kernels.embedding_rmsnorm(input_ids, w, n0.weight.data, n0.variance_epsilon)
    hidden_states = model.embed_tokens(input_ids)
    return hidden_states, None


def decode_model(model, hidden_states: torch.Tensor, positions: torch.Tensor, forward_batch, comm=None, first_normed=None) -> torch.Tensor:
    """The layer loop + final norm of LlamaModel.forward for a decode batch: returns norm(...) [M, hidden] (chunk-major at
    TP = 1: kernels.unblock).  `first_normed`: layer 0's input_layernorm of `hidden_states` when embed_and_norm() already made it."""
    layers = model.layers
    residual = hidden_states                     # the embedding output becomes the residual stream (llama.py:349-353)
    if comm is not None and not residual.is_contiguous():
        residual = residual.contiguous()
    x = first_normed if first_normed is not None else kernels.rmsnorm(hidden_states, layers[0].input_layernorm.weight.data,
                                                                        layers[0].input_layernorm.variance_epsilon)
    for i, layer in enumerate(layers):
        nxt = layers[i + 1].input_layernorm if i + 1 < len(layers) else model.norm
        step = decode_layer_moe if moe_block_of(layer) is not None else decode_layer
        x = step(layer, positions, x, forward_batch, residual, nxt, comm)
    return x


def model_fusable(model, hidden_states: torch.Tensor, forward_batch, comm=None) -> bool:
    """`comm`: the TP group's communicator when TP > 1 -- the [rows, hidden] partial sums of the row-parallel projections
    must be messages its kernels take (bf16, <= 64 MiB, hidden <= 16384 for the fused epilogue)."""
    if not (hidden_states.is_cuda and hidden_states.dtype == _BF16 and hidden_states.dim() == 2):
        return False
    return rows_fusable(model, hidden_states.shape[0], hidden_states.shape[1], forward_batch, comm)


def rows_fusable(model, rows: int, hidden: int, forward_batch, comm=None) -> bool:
    """model_fusable() from the SHAPE of the residual stream alone ([rows, hidden] bf16 on the device), so the hook can decide
    before it runs the embedding: at TP > 1 the vocab-parallel embedding ends in an all-reduce, which a refused batch would
    otherwise pay twice (once here, once inside the original forward)."""
    if comm is not None:
        nbytes = rows * hidden * 2
        if not (hidden <= 16384 and hidden % 8 == 0 and not comm.disabled and 0 < nbytes <= max(comm.max_bytes, comm.two_stage_bytes)):
            return False
    positions = getattr(forward_batch, "positions", None)
    if positions is not None and positions.dim() != 1:                 # (multimodal 3-D positions: not this form)
        return False
    mode = getattr(forward_batch, "forward_mode", None)
    if mode is None or not mode.is_decode():
        return False
    if not (len(model.layers) > 0 and all(layer_fusable(l, rows) for l in model.layers)):
        return False
    # a pool the kernels do not read (fp8_e5m2 rows, an unknown layout) must fall back BEFORE the first launch, not raise
    # in the middle of the forward
    try:
        from .layers.attention.hip_backend import pool_kernel_format

        pool_kernel_format(kv_pool_of(forward_batch), model.layers[0].self_attn.attn)
    except Exception:
        return False
    return True


def explain(model, forward_batch, hidden_states=None, input_embeds=None, pp_proxy_tensors=None) -> Optional[str]:
    """Why a forward of `model` (a LlamaModel / Qwen2Model) would NOT run the fused decode layer loop -- the first failing
    condition of the hook below, in words -- or None when it would.  For users and tests: the hook itself never raises."""
    if input_embeds is not None or pp_proxy_tensors is not None:
        return "input_embeds / pp_proxy_tensors given"
    if getattr(model, "layers_to_capture", None):
        return "aux hidden states are captured (layers_to_capture)"
    pp = getattr(model, "pp_group", None)
    if pp is not None and not (pp.is_first_rank and pp.is_last_rank):
        return "a pipeline stage"
    if getattr(model, "start_layer", 0) != 0 or getattr(model, "end_layer", len(model.layers)) != len(model.layers):
        return "start_layer / end_layer do not span the model"
    mode = getattr(forward_batch, "forward_mode", None)
    if mode is None or not mode.is_decode():
        return f"forward mode {mode} is not DECODE"
    tp, comm = _tp()
    if tp > 1 and comm is None:
        return f"TP = {tp} without an xGMI communicator on the TP group"
    positions = getattr(forward_batch, "positions", None)
    if positions is not None and positions.dim() != 1:
        return "positions are not one-dimensional"
    if hidden_states is not None and not (hidden_states.is_cuda and hidden_states.dtype == _BF16 and hidden_states.dim() == 2):
        return f"hidden states {hidden_states.dtype} {tuple(hidden_states.shape)} cuda={hidden_states.is_cuda}"
    rows = hidden_states.shape[0] if hidden_states is not None else int(forward_batch.batch_size)
    if len(model.layers) == 0:
        return "no layers"
    for i, layer in enumerate(model.layers):
        why = layer_unfusable_reason(layer, rows)
        if why is not None:
            return f"layer {i}: {why}"
    try:
        from .layers.attention.hip_backend import pool_kernel_format

        pool_kernel_format(kv_pool_of(forward_batch), model.layers[0].self_attn.attn)
    except Exception as e:                      # noqa: BLE001
        return f"KV pool format: {type(e).__name__}: {e}"
    return None


def llama_model_forward_hook(original, self, input_ids, positions, forward_batch, input_embeds=None, pp_proxy_tensors=None):
    """HookType.AROUND on LlamaModel.forward (llama.py:419-470) and Qwen2Model.forward (qwen2.py:396-448).  Decode batches
    of a single pipeline stage run the fused layer loop -- at TP > 1 with the TP group's xGMI all-reduce (add + RMSNorm
    in its epilogue) behind the row-parallel projections; anything else is the reference's own forward."""
    comm = None
    try:
        ok = _reference_model_applies(self, forward_batch, input_embeds, pp_proxy_tensors)
        if ok:
            tp, comm = _tp()
            ok = tp == 1 or comm is not None         # TP > 1 without the xGMI communicator: the reference's own layer loop
            comm = comm if tp > 1 else None
    except Exception:
        ok = False
    if ok:
        # decided from shapes BEFORE the embedding runs: a refused batch (more rows than the weight stream takes, a layer left to
        # the library GEMM) must not embed -- and, at TP > 1, all-reduce the embedding -- here and again in the original forward
        emb_w = getattr(getattr(self, "embed_tokens", None), "weight", None)
        rows = int(input_ids.shape[0]) if isinstance(input_ids, torch.Tensor) and input_ids.dim() == 1 else -1
        known = rows > 0 and isinstance(emb_w, torch.Tensor) and emb_w.dim() == 2
        if known:
            if emb_w.is_cuda and emb_w.dtype == _BF16 and rows_fusable(self, rows, int(emb_w.shape[1]), forward_batch, comm):
                hidden_states, normed0 = embed_and_norm(self, input_ids) if comm is None else (self.embed_tokens(input_ids), None)
                if model_fusable(self, hidden_states, forward_batch, comm):      # (what the embedding returned is what was assumed)
                    return kernels.unblock(decode_model(self, hidden_states, positions, forward_batch, comm, normed0))
                _say_once_why_not(self, forward_batch, hidden_states)
                return _finish_unfused(original, self, hidden_states, positions, forward_batch)
        else:
            # an embedding without a `.weight` to read the width from (not the reference's VocabParallelEmbedding): decided behind it
            hidden_states = self.embed_tokens(input_ids)
            if model_fusable(self, hidden_states, forward_batch, comm):
                return kernels.unblock(decode_model(self, hidden_states, positions, forward_batch, comm))
            _say_once_why_not(self, forward_batch, hidden_states)
            return original(self, input_ids, positions, forward_batch, input_embeds, pp_proxy_tensors)
        _say_once_why_not(self, forward_batch, None)
    return original(self, input_ids, positions, forward_batch, input_embeds, pp_proxy_tensors)


def _finish_unfused(original, model, hidden_states, positions, forward_batch):
    """The embedding has run but its output is not the [rows, hidden] bf16 tensor the decision assumed (never seen; kept so that the
    embedding is still not paid twice): hand the embeddings to the original forward as `input_embeds` (llama.py:433-437)."""
    return original(model, None, positions, forward_batch, hidden_states, None)


_SAID = set()


def _say_once_why_not(model, forward_batch, hidden_states) -> None:
    """A decode forward of a single-stage Llama-style model that does NOT take the fused loop is worth one log line per reason:
    the operator-by-operator path is correct but slower, and the cause (a quantised projection, an exotic rope, a pool the kernels
    do not read, a batch above the weight-stream's row limit ...) is otherwise invisible."""
    if len(_SAID) >= 16:
        return
    try:
        why = explain(model, forward_batch, hidden_states)
    except Exception as e:                          # noqa: BLE001 -- never let a diagnostic break a forward
        why = f"explain() raised {type(e).__name__}: {e}"
    if why is not None and why not in _SAID and len(_SAID) < 16:
        _SAID.add(why)
        import logging

        logging.getLogger("sglang_amd").info("decode forward of %s stays on the operator-by-operator path: %s", type(model).__name__, why)


def _reference_model_applies(model, forward_batch, input_embeds, pp_proxy_tensors) -> bool:
    if input_embeds is not None or pp_proxy_tensors is not None or getattr(model, "layers_to_capture", None):
        return False
    pp = getattr(model, "pp_group", None)
    if pp is not None and not (pp.is_first_rank and pp.is_last_rank):
        return False
    if getattr(model, "start_layer", 0) != 0 or getattr(model, "end_layer", len(model.layers)) != len(model.layers):
        return False
    mode = getattr(forward_batch, "forward_mode", None)
    return mode is not None and mode.is_decode()


def _tp():
    """(tensor-parallel degree, the TP group's xGMI communicator or None): the reference's TP group when running under it
    (tp_hooks.attach built its communicator inside GroupCoordinator.__init__), else this package's own process group."""
    from . import tp_hooks

    return tp_hooks.tp_communicator()


# model classes whose forward is the loop above (same signature, same attribute names; Mistral and the other Llama-style
# checkpoints are served by LlamaModel itself): llama.py:419-470, qwen2.py:396-448 (qkv bias: in the qkv combine)
HOOK_TARGETS = ("sglang.srt.models.llama.LlamaModel.forward", "sglang.srt.models.qwen2.Qwen2Model.forward",
                "sglang.srt.models.mixtral.MixtralModel.forward")         # mixtral.py:300-335: the same loop over sparse-MoE layers


def install(registry, hook_type_around) -> None:
    """plugin.load(): HookRegistry.register(target, hook, HookType.AROUND) for every model class of this form."""
    for target in HOOK_TARGETS:
        registry.register(target, llama_model_forward_hook, hook_type_around)
