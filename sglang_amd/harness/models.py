"""Synthetic-weight decoder models at the BASELINE configs' shapes, wired exactly like
the reference models: /root/reference/python/sglang/srt/models/llama.py:70-500
(LlamaMLP :70, LlamaAttention :138-223, LlamaDecoderLayer :283-370, LlamaModel :372,
LlamaForCausalLM :496), qwen2.py (qkv bias, tied embeddings), mixtral.py:60-118.

Only the operator calls differ: every non-GEMM op goes to the gfx950 kernels
through the layers in sglang_amd/layers; the projections of decode batches
(M <= 64) stream their weights through the gfx950 wstream GEMM (with the
residual-add + RMSNorm fused into its combine kernel at TP=1), prefill-sized
GEMMs go to hipBLASLt through torch (a plain library GEMM).  Weights are
random-init (there are no checkpoints offline): N(0, 0.02^2) bf16, norm weights 1
-- the `--load-format dummy` equivalent.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import fused_decode, kernels
from ..distributed import parallel_state as ps
from ..layers.activation import SiluAndMul
from ..layers.layernorm import RMSNorm
from ..layers.radix_attention import RadixAttention
from ..layers.rotary_embedding import FusedSetKVBufferArg, get_rope
from ..layers.sampler import LogitsProcessorOutput

BF = torch.bfloat16

# True: every operator runs through its own sglang.srt hook (RMSNorm.forward, RotaryEmbedding.forward with the
# fused KV store, SiluAndMul.forward, AttentionBackend.forward_*, the linear method) -- the path a drop-in under
# the unchanged reference model classes reaches.  False (default): TP=1 decode batches additionally fold the
# norms / rope / activation into the projections' combine kernels (needs the model-class patch of INTEGRATION.md).
import os as _os

OPERATOR_SURFACE_ONLY = _os.environ.get("SGLANG_AMD_OPERATOR_SURFACE", "0") == "1"


@dataclass
class ModelConfig:
    name: str
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    vocab_size: int
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Optional[Dict[str, Any]] = None
    max_position_embeddings: int = 8192
    attention_bias: bool = False
    tie_word_embeddings: bool = False
    num_local_experts: int = 0
    num_experts_per_tok: int = 0
    sliding_window: Optional[int] = None      # every layer attends [p - window, p] (mistral-style); None = full
    logit_cap: float = 0.0                    # attention logit soft cap (gemma-2 / grok style); 0 = off


_LLAMA3_ROPE = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                    original_max_position_embeddings=8192)

CONFIGS: Dict[str, ModelConfig] = {
    # SURVEY.md section 8 size table
    "llama-3-8b": ModelConfig("llama-3-8b", 4096, 14336, 32, 32, 8, 128, 128256, 1e-5, 500000.0, None, 8192),
    "llama-3-70b": ModelConfig("llama-3-70b", 8192, 28672, 80, 64, 8, 128, 128256, 1e-5, 500000.0, None, 8192),
    "qwen2.5-0.5b": ModelConfig("qwen2.5-0.5b", 896, 4864, 24, 14, 2, 64, 151936, 1e-6, 1000000.0, None, 32768,
                                attention_bias=True, tie_word_embeddings=True),
    "mixtral-8x7b": ModelConfig("mixtral-8x7b", 4096, 14336, 32, 32, 8, 128, 32000, 1e-5, 1000000.0, None, 32768,
                                num_local_experts=8, num_experts_per_tok=2),
    # LLaVA-1.6-7B's language model (vicuna-7B: MHA, rope theta 1e4); the vision tower is harness/llava.py ClipVisionConfig()
    "llava-1.6-7b": ModelConfig("llava-1.6-7b", 4096, 11008, 32, 32, 32, 128, 32064, 1e-5, 10000.0, None, 4096),
    # small shapes for tests / smoke
    "tiny-llama": ModelConfig("tiny-llama", 256, 512, 2, 8, 2, 64, 1024, 1e-5, 10000.0, None, 2048),
    "tiny-qwen": ModelConfig("tiny-qwen", 128, 384, 2, 4, 2, 64, 768, 1e-6, 10000.0, None, 2048,
                             attention_bias=True, tie_word_embeddings=True),
    "tiny-llama3-rope": ModelConfig("tiny-llama3-rope", 256, 512, 2, 4, 1, 128, 512, 1e-5, 500000.0, _LLAMA3_ROPE, 4096),
    "tiny-mixtral": ModelConfig("tiny-mixtral", 256, 512, 2, 8, 2, 64, 1024, 1e-5, 10000.0, None, 2048,
                                num_local_experts=8, num_experts_per_tok=2),
}


def _seed_of(name: str) -> int:
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def synth_weight(name: str, shape, init_device, std: float = 0.02) -> torch.Tensor:
    """Deterministic per-tensor init: the same (name, shape, init_device) gives the same bits."""
    g = torch.Generator(device=init_device)
    g.manual_seed(_seed_of(name))
    w = torch.randn(shape, generator=g, device=init_device, dtype=torch.float32)
    return (w * std).to(BF)


class Linear(nn.Module):
    """y = x W^T (+ b), W [out, in] bf16.  Decode batches (M <= 64) stream the weights through the
    gfx950 wstream GEMM; prefill-sized M uses hipBLASLt via torch (a plain library GEMM)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=False)
        self.bias = nn.Parameter(bias, requires_grad=False) if bias is not None else None

    def streams(self, x: torch.Tensor) -> bool:
        """x: row-major [M, K], or the chunk-major [K/128, M, 128] form the decode GEMMs hand to each other."""
        rows = x.shape[1] if x.dim() == 3 else x.shape[0]
        return (x.is_cuda and x.dim() in (2, 3) and x.stride(-1) == 1
                and kernels.wstream_preferred(rows, self.weight.shape[0], self.weight.shape[1]))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.streams(x):
            return kernels.wstream_gemm(x, self.weight.data, self.bias.data if self.bias is not None else None)
        return F.linear(kernels.unblock(x), self.weight, self.bias)

    def forward_all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        """RowParallelLinear.forward (linear.py): all_reduce(self(x)); prefill-sized inputs overlap the collective with
        the matmul piecewise (parallel_state.row_parallel_linear)."""
        if (ps.get_tensor_model_parallel_world_size() > 1 and self.bias is None and x.dim() == 2 and x.shape[0] >= ps.PIECEWISE_MIN_ROWS
                and not self.streams(x)):
            return ps.row_parallel_linear(x, self.weight)
        return ps.tensor_model_parallel_all_reduce(self.forward(x))

    def forward_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, norm: RMSNorm) -> torch.Tensor:
        """norm(self(x), residual) with the residual add + RMSNorm run by the GEMM's split-K combine
        kernel: `residual` is updated in place, the normed activations are returned -- chunk-major when the
        width allows, since only the next projection's GEMM reads them."""
        return kernels.wstream_gemm(x, self.weight.data, self.bias.data if self.bias is not None else None,
                                    epilogue="add_rmsnorm", residual=residual, norm_weight=norm.weight.data,
                                    eps=norm.variance_epsilon, out_blocked=self.weight.shape[0] % 128 == 0)


def _shard_rows(w: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    n = w.shape[0] // world
    return w[rank * n:(rank + 1) * n].contiguous()


def _shard_cols(w: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    n = w.shape[1] // world
    return w[:, rank * n:(rank + 1) * n].contiguous()


class LlamaMLP(nn.Module):
    def __init__(self, cfg: ModelConfig, prefix: str, init_device, device, tp_rank: int, tp_size: int):
        super().__init__()
        H, I = cfg.hidden_size, cfg.intermediate_size
        gate = _shard_rows(synth_weight(f"{prefix}.gate_proj", (I, H), init_device), tp_rank, tp_size)
        up = _shard_rows(synth_weight(f"{prefix}.up_proj", (I, H), init_device), tp_rank, tp_size)
        down = _shard_cols(synth_weight(f"{prefix}.down_proj", (H, I), init_device), tp_rank, tp_size)
        self.gate_up_proj = Linear(torch.cat([gate, up], 0).to(device))     # MergedColumnParallelLinear
        self.down_proj = Linear(down.to(device))                            # RowParallelLinear
        self.act_fn = SiluAndMul()

    def gate_up_act(self, x: torch.Tensor, out_blocked: bool = False) -> torch.Tensor:
        """act_fn(gate_up_proj(x)); decode batches get silu(gate) * up from the GEMM's own epilogue (chunk-major on
        request: the fused decode layer hands it to down_proj's GEMM only)."""
        if not OPERATOR_SURFACE_ONLY and self.gate_up_proj.streams(x) and self.gate_up_proj.bias is None:
            return kernels.wstream_gemm(x, self.gate_up_proj.weight.data, epilogue="silu_and_mul",
                                        out_blocked=out_blocked and self.gate_up_proj.weight.shape[0] % 256 == 0)
        return self.act_fn(self.gate_up_proj(x))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.down_proj.forward_all_reduce(self.gate_up_act(x))

    def forward_fused_norm(self, x: torch.Tensor, residual: torch.Tensor, next_norm: RMSNorm, tp_size: int = 1) -> torch.Tensor:
        """Decode: down_proj + residual add + the NEXT norm.  TP=1: one GEMM + combine pair; TP>1: the GEMM, then the
        one-shot xGMI all-reduce with the add + norm in its epilogue."""
        if tp_size > 1:
            y = self.down_proj(self.gate_up_act(x, out_blocked=True))
            return ps.tensor_model_parallel_all_reduce_add_rmsnorm(y, residual, next_norm.weight.data, next_norm.variance_epsilon)
        return self.down_proj.forward_add_rmsnorm(self.gate_up_act(x, out_blocked=True), residual, next_norm)


class LlamaAttention(nn.Module):
    def __init__(self, cfg: ModelConfig, layer_id: int, prefix: str, init_device, device, tp_rank: int, tp_size: int):
        super().__init__()
        H, D = cfg.hidden_size, cfg.head_dim
        self.total_q, self.total_kv = cfg.num_attention_heads, cfg.num_key_value_heads
        assert self.total_q % tp_size == 0
        self.num_heads = self.total_q // tp_size
        # KV heads are split, or replicated when there are fewer than TP ranks (llama.py:160-171)
        self.num_kv_heads = max(1, self.total_kv // tp_size)
        kv_rank = tp_rank * self.total_kv // tp_size if self.total_kv >= tp_size else tp_rank // (tp_size // self.total_kv)
        self.head_dim = D
        self.q_size, self.kv_size = self.num_heads * D, self.num_kv_heads * D
        wq = synth_weight(f"{prefix}.q_proj", (self.total_q * D, H), init_device)
        wk = synth_weight(f"{prefix}.k_proj", (self.total_kv * D, H), init_device)
        wv = synth_weight(f"{prefix}.v_proj", (self.total_kv * D, H), init_device)
        q = wq[tp_rank * self.q_size:(tp_rank + 1) * self.q_size]
        if self.total_kv >= tp_size:
            k = wk[tp_rank * self.kv_size:(tp_rank + 1) * self.kv_size]
            v = wv[tp_rank * self.kv_size:(tp_rank + 1) * self.kv_size]
        else:
            k = wk[kv_rank * D:(kv_rank + 1) * D]
            v = wv[kv_rank * D:(kv_rank + 1) * D]
        bias = None
        if cfg.attention_bias:
            bq = synth_weight(f"{prefix}.q_bias", (self.total_q * D,), init_device)
            bk = synth_weight(f"{prefix}.k_bias", (self.total_kv * D,), init_device)
            bv = synth_weight(f"{prefix}.v_bias", (self.total_kv * D,), init_device)
            if self.total_kv >= tp_size:
                bks = bk[tp_rank * self.kv_size:(tp_rank + 1) * self.kv_size]
                bvs = bv[tp_rank * self.kv_size:(tp_rank + 1) * self.kv_size]
            else:
                bks, bvs = bk[kv_rank * D:(kv_rank + 1) * D], bv[kv_rank * D:(kv_rank + 1) * D]
            bias = torch.cat([bq[tp_rank * self.q_size:(tp_rank + 1) * self.q_size], bks, bvs]).to(device)
        self.qkv_proj = Linear(torch.cat([q, k, v], 0).to(device), bias)    # QKVParallelLinear
        wo = synth_weight(f"{prefix}.o_proj", (H, self.total_q * D), init_device)
        self.o_proj = Linear(_shard_cols(wo, tp_rank, tp_size).to(device))  # RowParallelLinear
        self.rotary_emb = get_rope(D, D, cfg.max_position_embeddings, cfg.rope_theta, True, cfg.rope_scaling, BF, device)
        self.attn = RadixAttention(self.num_heads, D, D ** -0.5, self.num_kv_heads, layer_id, logit_cap=cfg.logit_cap,
                                   sliding_window_size=cfg.sliding_window if cfg.sliding_window is not None else -1)
        self.layer_id = layer_id

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor, forward_batch,
                fused_norm: Optional[Tuple[torch.Tensor, RMSNorm]] = None) -> torch.Tensor:
        """fused_norm = (residual, norm): TP=1 decode form, o_proj + residual add + norm in one GEMM +
        combine pair (returns the normed activations, residual updated in place)."""
        pool = forward_batch.token_to_kv_pool
        plain_pool = not getattr(pool, "is_fp8", False) and not getattr(pool, "use_hnd", False)
        streams = not OPERATOR_SURFACE_ONLY and self.qkv_proj.streams(hidden_states) and self.rotary_emb.is_neox_style
        if not plain_pool and not streams:
            # fp8 / HND pools: rope, then the backend stores the rows in the pool's own format (set_kv_buffer)
            qkv = self.qkv_proj(hidden_states)
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
            self.rotary_emb(positions, q, k)
            attn_output = self.attn(q, k, v, forward_batch, save_kv_cache=True)
            if fused_norm is not None:
                return self._o_proj_fused_norm(attn_output, fused_norm)
            return self.o_proj.forward_all_reduce(attn_output)
        if streams:
            # decode batch: qkv GEMM, rope and the KV-row store (in the pool's format) in one GEMM + combine pair
            q = kernels.wstream_qkv_rope(hidden_states, self.qkv_proj.weight.data,
                                         self.qkv_proj.bias.data if self.qkv_proj.bias is not None else None, positions,
                                         self.rotary_emb.cos_sin_cache, self.num_heads, self.num_kv_heads, self.head_dim,
                                         pool.get_key_buffer(self.layer_id), pool.get_value_buffer(self.layer_id),
                                         forward_batch.out_cache_loc,
                                         **({} if plain_pool else pool.kernel_format(self.attn)))
            attn_output = self.attn(q, None, None, forward_batch, save_kv_cache=False)
            if fused_norm is not None:
                return self._o_proj_fused_norm(attn_output, fused_norm)
            return self.o_proj.forward_all_reduce(attn_output)
        qkv = self.qkv_proj(hidden_states)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        # rope + KV-row scatter in one kernel (rotary_embedding/base.py:385-417), then attention reads the pool
        self.rotary_emb(positions, q, k, fused_set_kv_buffer_arg=FusedSetKVBufferArg(
            value=v, k_buffer=pool.get_key_buffer(self.layer_id), v_buffer=pool.get_value_buffer(self.layer_id),
            cache_loc=forward_batch.out_cache_loc))
        attn_output = self.attn(q, k, v, forward_batch, save_kv_cache=False)
        if fused_norm is not None:
            return self._o_proj_fused_norm(attn_output, fused_norm)
        return self.o_proj.forward_all_reduce(attn_output)

    def _o_proj_fused_norm(self, attn_output: torch.Tensor, fused_norm) -> torch.Tensor:
        residual, norm = fused_norm
        if ps.get_tensor_model_parallel_world_size() > 1:
            return ps.tensor_model_parallel_all_reduce_add_rmsnorm(self.o_proj(attn_output), residual, norm.weight.data,
                                                                   norm.variance_epsilon)
        return self.o_proj.forward_add_rmsnorm(attn_output, residual, norm)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, cfg: ModelConfig, layer_id: int, init_device, device, tp_rank: int, tp_size: int):
        super().__init__()
        p = f"model.layers.{layer_id}"
        self.self_attn = LlamaAttention(cfg, layer_id, f"{p}.self_attn", init_device, device, tp_rank, tp_size)
        if cfg.num_local_experts > 0:
            from .moe_block import SparseMoeBlock

            self.mlp = SparseMoeBlock(cfg, f"{p}.block_sparse_moe", init_device, device, tp_rank, tp_size)
        else:
            self.mlp = LlamaMLP(cfg, f"{p}.mlp", init_device, device, tp_rank, tp_size)
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, BF, device)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, BF, device)

    def forward(self, positions, hidden_states, forward_batch, residual) -> Tuple[torch.Tensor, torch.Tensor]:
        # llama.py:341-370
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, forward_batch)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual

    def fusable(self, x: torch.Tensor, tp_size: int) -> bool:
        """The decode form below needs no collective between a projection and the norm behind it."""
        if OPERATOR_SURFACE_ONLY or x.shape[1] > 16384:
            return False
        moe = fused_decode.moe_block_of(self) is not None
        if not moe and not isinstance(self.mlp, LlamaMLP):
            return False
        if tp_size > 1:     # the add + norm ride in the all-reduce's epilogue (one-shot or, for weak-scaled batches, two-stage)
            xg = ps.get_xgmi_all_reduce()
            return (xg is not None and x.is_cuda and x.dtype == BF and x.dim() == 2
                    and (xg.should_use(x) or xg.should_use_two_stage(x))
                    and (not moe or fused_decode.layer_fusable(self, x.shape[0])))     # (sparse-MoE layers: the fused attention half needs the weight stream)
        return x.is_cuda and fused_decode.layer_fusable(self, x.shape[0])     # the plug-in's own test (fused_decode.py)

    def forward_decode_fused(self, positions, normed: torch.Tensor, forward_batch, residual: torch.Tensor,
                             next_norm: RMSNorm) -> torch.Tensor:
        """Same arithmetic as forward() (llama.py:341-370) for a TP=1 decode batch, with every
        residual-add + RMSNorm executed by the preceding projection's combine kernel.  `normed` is
        this layer's input_layernorm output; returns next_norm's output, residual updated in place."""
        if fused_decode.moe_block_of(self) is not None:
            # sparse-MoE layer (Mixtral): fused attention half + the block's own gate / TopK / experts + add-norm (fused_decode.py),
            # what the AROUND hook on the reference's MixtralModel.forward runs
            xg = ps.get_xgmi_all_reduce() if ps.get_tensor_model_parallel_world_size() > 1 else None
            return fused_decode.decode_layer_moe(self, positions, normed, forward_batch, residual, next_norm, xg)
        if ps.get_tensor_model_parallel_world_size() == 1:
            # the very function plugin.load() hooks onto the reference's LlamaModel.forward: what bench.py times IS what
            # the plug-in delivers under unchanged reference model classes
            return fused_decode.decode_layer(self, positions, normed, forward_batch, residual, next_norm)
        xg = ps.get_xgmi_all_reduce()
        if xg is not None and fused_decode.layer_fusable(self, normed.shape[1] if normed.dim() == 3 else normed.shape[0]):
            # TP > 1, decode batches the weight-streaming GEMM takes: the plug-in's own layer function with the TP
            # group's communicator (what fused_decode.llama_model_forward_hook runs under the reference at TP > 1)
            return fused_decode.decode_layer(self, positions, normed, forward_batch, residual, next_norm, xg)
        x = self.self_attn(positions, normed, forward_batch, fused_norm=(residual, self.post_attention_layernorm))
        return self.mlp.forward_fused_norm(x, residual, next_norm, ps.get_tensor_model_parallel_world_size())


class CausalLM(nn.Module):
    """LlamaForCausalLM / Qwen2ForCausalLM / MixtralForCausalLM on one TP rank."""

    def __init__(self, cfg: ModelConfig, device, init_device=None, tp_rank: int = 0, tp_size: int = 1):
        super().__init__()
        self.config = cfg
        self.device = device
        self.tp_rank, self.tp_size = tp_rank, tp_size
        init_device = init_device if init_device is not None else device
        # replicated embedding (the reference shards it over vocab + all-reduce; at
        # T x hidden the replicated gather is cheaper than a collective on xGMI)
        self.embed_tokens = nn.Parameter(
            synth_weight("model.embed_tokens", (cfg.vocab_size, cfg.hidden_size), init_device).to(device),
            requires_grad=False)
        self.layers = nn.ModuleList(
            [LlamaDecoderLayer(cfg, i, init_device, device, tp_rank, tp_size) for i in range(cfg.num_hidden_layers)])
        self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, BF, device)
        if cfg.tie_word_embeddings:
            head = self.embed_tokens.data
        else:
            head = synth_weight("lm_head", (cfg.vocab_size, cfg.hidden_size), init_device).to(device)
        assert cfg.vocab_size % tp_size == 0
        n = cfg.vocab_size // tp_size
        self.lm_head = nn.Parameter(head[tp_rank * n:(tp_rank + 1) * n].contiguous() if tp_size > 1 else head,
                                    requires_grad=False)                    # vocab-parallel head

    @property
    def num_attention_heads_per_rank(self) -> int:
        return self.layers[0].self_attn.num_heads

    @property
    def num_kv_heads_per_rank(self) -> int:
        return self.layers[0].self_attn.num_kv_heads

    def forward_hidden(self, input_ids: torch.Tensor, positions: torch.Tensor, forward_batch) -> torch.Tensor:
        embeds = getattr(forward_batch, "input_embeds", None)
        x = None
        if (embeds is None and forward_batch.forward_mode.is_decode() and input_ids.is_cuda and input_ids.dtype == torch.int64
                and input_ids.dim() == 1 and self.embed_tokens.dtype == BF and not OPERATOR_SURFACE_ONLY):
            # decode step: the embedding lookup and the first layer's input norm in one launch (fused_decode.embed_and_norm does the
            # same on a reference model)
            n0 = self.layers[0].input_layernorm
            hidden_states, x = kernels.embedding_rmsnorm(input_ids, self.embed_tokens.data, n0.weight.data, n0.variance_epsilon)
        else:
            hidden_states = embeds if embeds is not None else F.embedding(input_ids, self.embed_tokens)
        if all(layer.fusable(hidden_states, self.tp_size) for layer in self.layers):
            residual = hidden_states                     # the embedding output becomes the residual stream
            if x is None:
                x = self.layers[0].input_layernorm(hidden_states)
            for i, layer in enumerate(self.layers):
                nxt = self.layers[i + 1].input_layernorm if i + 1 < len(self.layers) else self.norm
                x = layer.forward_decode_fused(positions, x, forward_batch, residual, nxt)
            return x
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, forward_batch, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states

    def compute_logits(self, hidden_states: torch.Tensor, forward_batch) -> LogitsProcessorOutput:
        """logits_processor.py:652-700: last token of every request, vocab-parallel head + all-gather."""
        if getattr(forward_batch.forward_mode, "is_target_verify", lambda: False)():
            hidden_states = kernels.unblock(hidden_states)          # every draft token is scored (logits_processor.py: verify keeps all rows)
        elif forward_batch.forward_mode.is_extend():
            last = torch.cumsum(forward_batch.extend_seq_lens, dim=0, dtype=torch.int64) - 1
            hidden_states = kernels.unblock(hidden_states)[last]
        rows = hidden_states.shape[1] if hidden_states.dim() == 3 else hidden_states.shape[0]
        if hidden_states.dim() == 3 and not kernels.wstream_preferred(rows, *self.lm_head.shape):
            hidden_states = kernels.unblock(hidden_states)
        if hidden_states.is_cuda and kernels.wstream_preferred(rows, *self.lm_head.shape):
            logits = kernels.wstream_gemm(hidden_states, self.lm_head.data)
        else:
            logits = F.linear(hidden_states, self.lm_head)
        logits = ps.tensor_model_parallel_all_gather(logits, dim=-1)
        # kept in the model dtype: the greedy sampler arg-maxes the bf16 logits directly, the sampling path
        # widens them to fp32 (logits_processor.py returns fp32; bf16 -> fp32 is exact, so nothing changes)
        return LogitsProcessorOutput(next_token_logits=logits)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, forward_batch) -> LogitsProcessorOutput:
        return self.compute_logits(self.forward_hidden(input_ids, positions, forward_batch), forward_batch)
