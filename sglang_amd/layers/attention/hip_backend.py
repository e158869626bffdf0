"""HipAttnBackend: the gfx950 attention backend behind SGLang's AttentionBackend surface.

Takes the place of TritonAttnBackend / TorchNativeAttnBackend
(/root/reference/python/sglang/srt/layers/attention/triton_backend.py:136-1012,
torch_native_backend.py:19-398).  Differences that matter on MI355X:

  * no kv_indptr / kv_indices gather per step: the HIP kernels walk
    req_to_token[req_pool_indices[b], :] directly, so the per-step metadata is
    just an int32 view of seq_lens (decode) or one tiny qo_indptr (extend);
    the decode path therefore needs NO out-of-graph refresh besides the
    ForwardBatch buffers the graph runner already copies -- nothing
    data-dependent happens on the host inside or around a hipGraph replay;
  * split-KV count is a static function of (batch bucket, kv heads, context
    bound), fixed at capture; empty splits exit immediately;
  * fp32 split workspaces are allocated once (`init_cuda_graph_state`) and
    owned by the backend, never by the kernels.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from ... import kernels
from ...mem_cache.memory_pool import KVWriteLoc, _host_scale
from .base_attn_backend import AttentionBackend


def _kv_write_loc(pool, loc):
    """The POOL's own KVWriteLoc type: the reference's for a reference pool (memory_pool.py:1550-1575), ours for ours --
    decided by whose class the pool is, not by whether `sglang` happens to be importable in the process."""
    if any(c.__module__.startswith("sglang.") for c in type(pool).__mro__):      # incl. mem_hooks' subclass of the reference pool
        try:
            from sglang.srt.mem_cache.memory_pool import KVWriteLoc as Ref

            return Ref(loc, None)
        except Exception:
            pass
    return KVWriteLoc(loc, None)


@dataclass
class _Meta:
    seq_lens_i32: torch.Tensor
    num_splits: int = 1
    ws_acc: Optional[torch.Tensor] = None
    ws_ml: Optional[torch.Tensor] = None
    qo_indptr: Optional[torch.Tensor] = None
    prefix_lens_i32: Optional[torch.Tensor] = None
    max_extend_len: int = 0
    cascade: Optional["kernels.CascadeWorkspace"] = None   # shared-prefix decode plan + split slots
    mask_indptr: Optional[torch.Tensor] = None             # TARGET_VERIFY: offsets of the requests' blocks in the flat mask
    verify_mask: Optional[torch.Tensor] = None             # TARGET_VERIFY: the persistent copy of this step's flat tree mask


def choose_num_splits(batch: int, num_kv_heads: int, group: int, max_len: int, target_blocks: int = 512) -> int:
    """Static split-KV choice: enough workgroups to fill 256 CUs twice, but never
    chunks shorter than 2 x the kernel's minimum chunk."""
    head_blocks = max(1, (group + 7) // 8) if group > 4 else 1
    blocks = max(1, batch * num_kv_heads * head_blocks)
    want = (target_blocks + blocks - 1) // blocks
    cap = max(1, max_len // (2 * kernels.decode_min_chunk()))
    return max(1, min(want, cap, 64))


def runner_head_dims(model_runner):
    """(q heads, kv heads, head dim) of this TP rank.  The harness ModelRunner carries them as fields; the
    reference ModelRunner (model_executor/model_runner.py) carries `model_config` (configs/model_config.py:1106-1196
    get_num_attention_heads / get_num_kv_heads) and the attention TP size (triton_backend.py:221-227)."""
    if hasattr(model_runner, "num_attention_heads_per_rank"):
        return (model_runner.num_attention_heads_per_rank, model_runner.num_kv_heads_per_rank, model_runner.head_dim)
    mc = model_runner.model_config
    tp = None
    try:
        from sglang.srt.runtime_context import get_parallel      # triton_backend.py:219-227

        tp = get_parallel().attn_tp_size
    except Exception:
        pass
    if tp is None:
        tp = getattr(model_runner, "attn_tp_size", None) or getattr(model_runner, "tp_size", 1)
    if hasattr(mc, "get_num_attention_heads"):
        hq = mc.get_num_attention_heads(tp)
    else:
        hq = max(1, mc.num_attention_heads // tp)
    hkv = mc.get_num_kv_heads(tp) if hasattr(mc, "get_num_kv_heads") else max(1, mc.num_key_value_heads // tp)
    pool = model_runner.token_to_kv_pool
    kbuf = pool.get_key_buffer(getattr(pool, "start_layer", 0) or 0)
    if int(kbuf.shape[-2]) != hkv:
        raise RuntimeError(f"KV pool holds {int(kbuf.shape[-2])} kv heads per token, the model config says {hkv} per rank")
    return hq, hkv, int(kbuf.shape[-1])


def pool_kernel_format(pool, layer=None) -> dict:
    """(kv_fp8, k_scale, v_scale, page_size, hnd) of ANY MHA pool for the attention / store kernels: this package's
    pool, or the reference's MHATokenToKVPool under the drop-in path (memory_pool.py:1636-1653 `dtype` / `store_dtype`,
    :1816 `use_hnd`, `page_size`).  The format is read off the pool's own attributes, never assumed; fp8 scales are the
    layer's host floats (radix_attention.py:129-130 k_scale_float / v_scale_float, set when a checkpoint carries
    scales) -- never float(device tensor), which would synchronise and break graph capture."""
    dtype = getattr(pool, "dtype", torch.bfloat16)
    if dtype in (getattr(torch, "float8_e5m2", None), getattr(torch, "float8_e4m3fnuz", None)):
        raise NotImplementedError(f"KV pool dtype {dtype}: the gfx950 kernels read OCP e4m3 (float8_e4m3fn) or bf16 rows")
    fp8 = dtype == torch.float8_e4m3fn
    ks = vs = 1.0
    if fp8 and layer is not None:
        ks, vs = _host_scale(layer, "k_scale"), _host_scale(layer, "v_scale")
    page = int(getattr(pool, "page_size", 1) or 1)
    return dict(kv_fp8=fp8, k_scale=ks, v_scale=vs, page_size=page, hnd=bool(getattr(pool, "use_hnd", False)) and page > 1)


def cascade_wanted(model_runner) -> bool:
    """Shared-prefix (cascade) decode plan or the plain paged decode kernel, decided once per server (the choice is baked into the
    captured decode graphs): an explicit `model_runner.enable_cascade_attention`, else `SGLANG_AMD_CASCADE=0 / 1`, else ON -- also for
    a server without a radix cache: on batches with nothing shared the plan's private items cost what the plain kernel costs
    (the reference's bench_one_batch, Llama-3-8B, `profiles/r05_exp2_cascade_policy_unshared.json`: 64 x 1024 6.09 against 5.94 ms
    per step, 16 x 4096 5.41 against 5.72 ms), so there is no rule worth keying on `disable_radix_cache`."""
    wanted = getattr(model_runner, "enable_cascade_attention", None)
    if wanted is None:
        env = os.environ.get("SGLANG_AMD_CASCADE", "")
        return env != "0"
    return bool(wanted)


class HipAttnBackend(AttentionBackend):
    needs_cpu_seq_lens = True
    # qo_indptr / lens are sized per forward, never preallocated at (req pool + 1) (base_attn_backend.py:117-122)
    extend_dummy_seqs_capped_by_req_pool = False

    def __init__(self, model_runner):
        """Reads what the reference backends read of a ModelRunner (triton_backend.py:145-206,
        torch_native_backend.py:20-33): `device`, `req_to_token_pool`, `token_to_kv_pool`, and for the head
        counts `model_config` + the attention TP size -- or the harness runner's per-rank fields."""
        super().__init__()
        self.device = model_runner.device
        self.req_to_token_pool = model_runner.req_to_token_pool
        self.token_to_kv_pool = model_runner.token_to_kv_pool
        self.num_q_heads, self.num_kv_heads, self.head_dim = runner_head_dims(model_runner)
        # refused when the server builds its backend, not at the first prefill: the prefill kernel covers head dims 64 / 128
        # (extend_attention.hip), activations are bf16, pool rows bf16 or OCP e4m3
        if self.head_dim not in (64, 128):
            raise NotImplementedError(f"hip_mi355x attention backend: head_dim {self.head_dim} (the gfx950 kernels cover 64 and 128); "
                                      f"run this model with another --attention-backend")
        act_dtype = getattr(model_runner, "dtype", None)
        if isinstance(act_dtype, torch.dtype) and act_dtype != torch.bfloat16:
            raise NotImplementedError(f"hip_mi355x attention backend: model dtype {act_dtype} (the gfx950 kernels compute in bf16: "
                                      f"--dtype bfloat16, or another --attention-backend)")
        pool_dtype = getattr(self.token_to_kv_pool, "dtype", torch.bfloat16)
        if pool_dtype not in (torch.bfloat16, torch.float8_e4m3fn):
            raise NotImplementedError(f"hip_mi355x attention backend: KV pool dtype {pool_dtype} (bf16 or float8_e4m3fn rows)")
        self.max_context_len = int(self.req_to_token_pool.req_to_token.shape[1])
        self.sliding_window_size = getattr(model_runner, "sliding_window_size", None)
        self.forward_metadata: Optional[_Meta] = None
        self._graph_ws = {}
        self._cascade_ws = None
        self._cascade_ws_eager = None
        self._cascade_in_graph = False
        self._seq_i32, self._seq_src, self._seq_i32_in_graph = None, None, False
        self._verify_states, self._verify_nd = {}, 0
        self._verify_mask, self._verify_mask_in_graph = None, False      # the flat tree mask captured verify graphs read (see _verify_mask_buffer)
        self.debug_flags = 0
        # RadixAttention batches share KV prefixes; the cascade decode path reads a shared prefix once per group (see cascade_wanted)
        self.enable_cascade = cascade_wanted(model_runner) and self.head_dim in (64, 128)
        # the shared-prefix kernel reads every pool format (bf16 / fp8 rows, token-major / paged head-major); layers
        # with a sliding window or a logit cap take the plain paged kernel (forward_decode)
        fmt = pool_kernel_format(self.token_to_kv_pool, None)
        self.plain_pool = not fmt["kv_fp8"] and not fmt["hnd"]

    # ------------------------------------------------------------------ metadata
    def _workspace(self, batch: int, splits: int):
        key = (batch, splits)
        ws = self._graph_ws.get(key)
        if ws is None:
            ws = kernels.decode_workspace(batch, self.num_q_heads, self.head_dim, splits, self.device)
            self._graph_ws[key] = ws
        return ws

    def _cascade_workspace(self, batch: int, in_capture: bool = False):
        """ONE workspace, sized for the largest batch seen (the graph runner asks for max_bs first): the plan and
        the slot layout are addressed with the actual batch size of the step, and replays are stream-serialised,
        so every bucket and every eager batch can share it.  It only ever grows before graphs exist."""
        ws = self._cascade_ws
        if ws is None or ws.max_batch < batch:
            if ws is not None and self._cascade_in_graph:
                # A batch larger than every captured bucket can only be an EAGER forward (replays never exceed max_bs;
                # decode_cuda_graph_runner.py:683-687 `cuda_graph_bs <= self.max_bs` sends it to the eager runner): it gets a
                # workspace of its own, the captured graphs keep theirs.
                if in_capture:
                    raise RuntimeError(f"cascade workspace holds {ws.max_batch} requests and is referenced by captured "
                                       f"graphs; capturing a batch of {batch} needs init_cuda_graph_state(max_bs >= {batch})")
                ew = self._cascade_ws_eager
                if ew is None or ew.max_batch < batch:
                    ew = self._cascade_ws_eager = kernels.CascadeWorkspace(batch, self.num_q_heads, self.head_dim,
                                                                           self.max_context_len, self.device)
                return ew
            ws = self._cascade_ws = kernels.CascadeWorkspace(batch, self.num_q_heads, self.head_dim,
                                                             self.max_context_len, self.device)
        return ws

    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int):
        """Pre-allocate the split workspaces for the largest bucket (reference :159)."""
        splits = choose_num_splits(1, self.num_kv_heads, self.num_q_heads // self.num_kv_heads, self.max_context_len)
        self._workspace(max_bs, max(splits, 1))
        if self._seq_i32 is None or self._seq_i32.numel() < max_bs:
            self._seq_i32 = torch.zeros(max(max_bs, 256), dtype=torch.int32, device=self.device)
        if self.enable_cascade and max_bs >= 2:
            self._cascade_workspace(min(max_bs, 1024))
        if max_num_tokens > max_bs and self._verify_mask is None:
            # speculative decoding: the graphs are TARGET_VERIFY forwards of max_num_tokens / max_bs draft tokens per request; their
            # masks hold draft_tokens x (context + draft_tokens) entries per request (triton_backend.py sizes its buffer the same way)
            self._verify_mask = torch.zeros(int(max_num_tokens) * self.max_context_len, dtype=torch.uint8, device=self.device)

    def get_cuda_graph_seq_len_fill_value(self):
        return 1

    def _seq_lens_i32(self, fb, in_capture: bool = False):
        """int32 view of the batch's seq_lens for the kernels.  The reference's batches carry int64 seq_lens
        (schedule_batch.py): the conversion then goes through a PERSISTENT buffer that init_forward_metadata_in_graph
        fills -- inside the decode graph, so a replay converts the step's own lengths (a `.to(int32)` here, outside the
        graph, would leave the captured kernels reading the capture-time copy).  A batch LARGER than the buffer the captured
        graphs reference can only be an eager forward (`--cuda-graph-max-bs 256` with 300 running requests: the reference's
        `can_run_graph` refuses it and the eager runner calls `init_forward_metadata_out_graph`): it is converted on the spot."""
        if fb.seq_lens.dtype == torch.int32:
            return fb.seq_lens, None
        bs = fb.seq_lens.numel()
        buf = self._seq_i32
        if buf is None or buf.numel() < bs:
            if buf is not None and self._seq_i32_in_graph:
                if in_capture:
                    raise RuntimeError(f"seq_lens buffer holds {buf.numel()} requests and is referenced by captured graphs; "
                                       f"capturing a batch of {bs} needs init_cuda_graph_state(max_bs >= {bs})")
                return fb.seq_lens.to(torch.int32), None
            buf = self._seq_i32 = torch.zeros(max(bs, 256), dtype=torch.int32, device=self.device)
        return buf[:bs], fb.seq_lens

    def _verify_mask_buffer(self, mask: torch.Tensor, in_capture: bool):
        """The flat tree mask of a TARGET_VERIFY forward, in a buffer that PERSISTS.  Every verify step builds a fresh mask tensor
        (ngram_worker.py:352-369, eagle builds its own): a captured graph would keep reading the tensor of the step it was captured
        on.  Like the reference's backends (triton_backend.py:559-566 `cuda_graph_custom_mask[: n] = spec_info.custom_mask`) the
        step's mask is copied -- here, outside the graph, before every replay and every eager forward -- into one buffer sized for
        the largest verify batch (tokens x context), and the kernels read THAT.  Round 5: found by running NGRAM speculative
        decoding under the reference's scheduler -- eager verify steps reproduced greedy decoding, replayed ones did not."""
        m8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
        m8 = m8.reshape(-1)
        buf = self._verify_mask
        if buf is None or buf.numel() < m8.numel():
            if buf is not None and self._verify_mask_in_graph:
                if in_capture:
                    raise RuntimeError(f"verify-mask buffer holds {buf.numel()} entries and is referenced by captured graphs; capturing a "
                                       f"mask of {m8.numel()} needs init_cuda_graph_state(max_num_tokens) to cover it")
                return m8                                      # an eager verify batch beyond the captured sizes: its own tensor
            buf = self._verify_mask = torch.zeros(max(m8.numel(), 1), dtype=torch.uint8, device=self.device)
        buf[: m8.numel()].copy_(m8, non_blocking=True)
        self._verify_mask_in_graph |= in_capture
        return buf

    def _verify_state(self, bs: int, nd: int):
        st = self._verify_states.get((bs, nd))
        if st is None:
            dev = self.device
            st = self._verify_states[(bs, nd)] = dict(
                qo=torch.arange(0, (bs + 1) * nd, nd, dtype=torch.int32, device=dev), pre=torch.zeros(bs, dtype=torch.int32, device=dev),
                kv=torch.zeros(bs, dtype=torch.int32, device=dev), kv64=torch.zeros(bs, dtype=torch.int64, device=dev),
                mip=torch.zeros(bs + 1, dtype=torch.int64, device=dev))
        return st

    def init_forward_metadata_out_graph(self, forward_batch, in_capture: bool = False):
        fb = forward_batch
        if fb.forward_mode.is_decode():
            seq_i32, self._seq_src = self._seq_lens_i32(fb, in_capture)
            self._seq_i32_in_graph |= in_capture and self._seq_src is not None
        else:
            seq_i32, self._seq_src = (fb.seq_lens if fb.seq_lens.dtype == torch.int32 else fb.seq_lens.to(torch.int32)), None
        if fb.forward_mode.is_decode():
            if in_capture:
                max_len = self.max_context_len       # the graph must be valid for any later length
            else:
                max_len = int(fb.seq_lens_cpu.max()) if fb.seq_lens_cpu is not None else self.max_context_len
            if self.enable_cascade and 2 <= fb.batch_size <= 1024:
                self.forward_metadata = _Meta(seq_i32, cascade=self._cascade_workspace(fb.batch_size, in_capture))
                self._cascade_in_graph |= in_capture
                return
            splits = choose_num_splits(fb.batch_size, self.num_kv_heads, self.num_q_heads // self.num_kv_heads, max_len)
            ws = self._workspace(fb.batch_size, splits) if splits > 1 else (None, None)
            self.forward_metadata = _Meta(seq_i32, splits, ws[0], ws[1])
        elif getattr(fb.forward_mode, "is_target_verify", lambda: False)():
            # triton_backend.py:860-919: every request extends by `draft_token_num` tokens over its whole context
            # (forward_batch.seq_lens = the context BEFORE the draft tokens), qo_indptr = arange * draft_token_num, the
            # mask block of request b holds draft_token_num x (seq_len + draft_token_num) entries.  Every tensor lives in a
            # per-(batch, draft tokens) state that persists, and the length-dependent ones are filled by
            # init_forward_metadata_in_graph: a captured verify graph then recomputes them from the step's own seq_lens.
            nd = int(getattr(fb.spec_info, "draft_token_num"))
            st = self._verify_state(fb.batch_size, nd)
            self._seq_src = fb.seq_lens
            self.forward_metadata = _Meta(st["kv"], qo_indptr=st["qo"], prefix_lens_i32=st["pre"], max_extend_len=nd, mask_indptr=st["mip"])
            self._verify_nd = nd
            cm = getattr(fb.spec_info, "custom_mask", None)
            self.forward_metadata.verify_mask = self._verify_mask_buffer(cm, in_capture) if isinstance(cm, torch.Tensor) else None
        elif fb.forward_mode.is_extend():
            ext = fb.extend_seq_lens_cpu
            qo = torch.zeros(fb.batch_size + 1, dtype=torch.int32)
            qo[1:] = torch.cumsum(torch.tensor(ext, dtype=torch.int32), 0)
            self.forward_metadata = _Meta(seq_i32, qo_indptr=qo.to(self.device, non_blocking=True),
                                          prefix_lens_i32=fb.extend_prefix_lens.to(torch.int32),
                                          max_extend_len=max(ext) if ext else 0)
        else:
            self.forward_metadata = None

    def init_forward_metadata_in_graph(self, forward_batch):
        """Device-only, static-shape work recorded into the decode graph: the shared-prefix plan of
        this step (one launch per step, reused by every layer).  The plain decode kernel needs nothing."""
        m = self.forward_metadata
        if m is not None and getattr(forward_batch.forward_mode, "is_target_verify", lambda: False)():
            st = self._verify_state(forward_batch.batch_size, self._verify_nd)
            st["pre"].copy_(forward_batch.seq_lens)              # context before the draft tokens
            torch.add(st["pre"], self._verify_nd, out=st["kv"])  # context incl. the draft tokens
            st["kv64"].copy_(st["kv"])
            st["kv64"].mul_(self._verify_nd)                     # mask entries per request
            torch.cumsum(st["kv64"], 0, out=st["mip"][1:])
            return
        if m is not None and self._seq_src is not None and forward_batch.forward_mode.is_decode():
            m.seq_lens_i32.copy_(forward_batch.seq_lens)         # int64 -> int32, recorded into the graph
        if m is not None and m.cascade is not None and forward_batch.forward_mode.is_decode():
            kernels.cascade_plan(m.cascade, self.req_to_token_pool.req_to_token, self._pool_idx(forward_batch),
                                 m.seq_lens_i32, self.num_q_heads, self.num_kv_heads)

    # ------------------------------------------------------------------ forward
    def _save_kv(self, layer, forward_batch, k, v):
        self.token_to_kv_pool.set_kv_buffer(layer, _kv_write_loc(self.token_to_kv_pool, forward_batch.out_cache_loc), k, v)

    @staticmethod
    def _pool_idx(forward_batch):
        idx = forward_batch.req_pool_indices
        return idx if idx.dtype == torch.int64 else idx.to(torch.int64)

    def _layer_options(self, layer):
        """Pool format + the per-layer attention switches the reference kernels honour (radix_attention.py:115-148:
        logit_cap, sliding_window_size, k_scale / v_scale)."""
        opt = pool_kernel_format(self.token_to_kv_pool, layer)
        win = getattr(layer, "sliding_window_size", -1)
        if win is not None and win > -1:
            opt["sliding_window"] = int(win)
        cap = getattr(layer, "logit_cap", 0.0) or 0.0
        if cap > 0:
            opt["logit_cap"] = float(cap)
        if not (opt.get("kv_fp8") or opt.get("hnd") or "sliding_window" in opt or "logit_cap" in opt):
            return {}
        return opt

    @staticmethod
    def _refuse_unsupported(layer, kwargs) -> None:
        """What a model may ask of an attention backend through RadixAttention.forward's **kwargs / layer attributes
        (radix_attention.py:150-159; triton_backend.py forward_extend / forward_decode honour them) and these kernels do not
        compute: refused by name instead of answered without it."""
        for name in ("sinks", "k_rope", "q_rope", "idx_q"):
            if kwargs.get(name) is not None:
                raise NotImplementedError(f"hip_mi355x attention backend: `{name}` (attention sinks / MLA rope split / sparse index "
                                          f"attention) is not implemented; run this model with another --attention-backend")
        if (getattr(layer, "xai_temperature_len", -1) or -1) > 0:
            raise NotImplementedError("hip_mi355x attention backend: xai_temperature_len (Grok's length-dependent temperature) is not implemented")
        if getattr(layer, "is_cross_attention", False):
            raise NotImplementedError("hip_mi355x attention backend: cross attention (encoder KV rows) is not implemented")
        if layer.qk_head_dim != layer.v_head_dim:
            raise NotImplementedError(f"hip_mi355x attention backend: qk_head_dim {layer.qk_head_dim} != v_head_dim {layer.v_head_dim} (MLA layouts)")

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True, **kwargs):
        self._refuse_unsupported(layer, kwargs)
        if save_kv_cache and k is not None and v is not None:
            self._save_kv(layer, forward_batch, k, v)
        m = self.forward_metadata
        q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
        o = torch.empty(q3.shape, dtype=q3.dtype, device=q3.device)
        # (compared by VALUE: under sglang `layer` is the reference's RadixAttention and attn_type a member of the reference's
        # own AttentionType enum, radix_attention.py:58-66 -- another class than this package's look-alike)
        causal = getattr(layer.attn_type, "value", layer.attn_type) != "encoder_only"
        opt = self._layer_options(layer)
        # speculative-decoding verify / tree attention (triton_backend.py:860-919): spec_info carries the flat mask
        spec = getattr(forward_batch, "spec_info", None)
        if spec is not None and getattr(spec, "custom_mask", None) is not None:
            # (a verify forward reads the persistent copy init_forward_metadata_out_graph made: the tensor on spec_info is this
            # step's own and would be baked into a captured graph)
            opt["custom_mask"] = m.verify_mask if getattr(m, "verify_mask", None) is not None else spec.custom_mask
            opt["mask_indptr"] = m.mask_indptr if m.mask_indptr is not None else self._mask_indptr(forward_batch, m)
            # the reference's verify call leaves skip_prefix_custom_mask at its default: the prefix is fully visible,
            # the mask decides among the draft tokens only (extend_attention.py:774)
            opt["skip_prefix_custom_mask"] = getattr(forward_batch.forward_mode, "is_target_verify", lambda: False)()
        kernels.extend_attention(q3, o, self.token_to_kv_pool.get_key_buffer(layer.layer_id),
                                 self.token_to_kv_pool.get_value_buffer(layer.layer_id),
                                 self.req_to_token_pool.req_to_token, self._pool_idx(forward_batch), m.seq_lens_i32,
                                 m.prefix_lens_i32, m.qo_indptr, m.max_extend_len, layer.scaling, causal, **opt)
        return o.view(-1, layer.tp_q_head_num * layer.v_head_dim)

    @staticmethod
    def _mask_indptr(forward_batch, m):
        """Offsets of every request's [extend_len, kv_len] block in the flat mask (triton_backend.py:880-895:
        mask_indptr = cumsum(extend_len * kv_len))."""
        ext = m.qo_indptr[1:].to(torch.int64) - m.qo_indptr[:-1].to(torch.int64)
        sizes = ext * m.seq_lens_i32.to(torch.int64)
        out = torch.zeros(sizes.numel() + 1, dtype=torch.int64, device=sizes.device)
        out[1:] = torch.cumsum(sizes, 0)
        return out

    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True, **kwargs):
        self._refuse_unsupported(layer, kwargs)
        if save_kv_cache and k is not None and v is not None:
            self._save_kv(layer, forward_batch, k, v)
        m = self.forward_metadata
        q3 = q.reshape(-1, layer.tp_q_head_num, layer.qk_head_dim)
        o = torch.empty_like(q3)
        opt = self._layer_options(layer)
        if m.cascade is not None and "sliding_window" not in opt and "logit_cap" not in opt:
            kernels.cascade_decode_attention(m.cascade, q3, self.token_to_kv_pool.get_key_buffer(layer.layer_id),
                                             self.token_to_kv_pool.get_value_buffer(layer.layer_id), o,
                                             self.req_to_token_pool.req_to_token, self._pool_idx(forward_batch),
                                             m.seq_lens_i32, layer.scaling, **opt)      # opt: the pool format, if any
            return o.view(-1, layer.tp_q_head_num * layer.v_head_dim)
        kernels.decode_attention(q3, self.token_to_kv_pool.get_key_buffer(layer.layer_id),
                                 self.token_to_kv_pool.get_value_buffer(layer.layer_id), o,
                                 self.req_to_token_pool.req_to_token, self._pool_idx(forward_batch), m.seq_lens_i32,
                                 layer.scaling, m.num_splits, m.ws_acc, m.ws_ml, flags=self.debug_flags, **opt)
        return o.view(-1, layer.tp_q_head_num * layer.v_head_dim)

    def forward_mixed(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        """ForwardMode.MIXED (forward_batch_info.py:100-110): chunked-prefill extends and running decodes in ONE
        batch, the decodes expressed as extends of one token over their cached prefix -- exactly what the extend
        kernel computes (the reference routes MIXED to forward_extend on every device but the NPU, :216-258)."""
        return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache)

    def support_triton(self) -> bool:
        return False
