"""ModelRunner + a minimal scheduler loop that drives the hot path the way the
reference scheduler does (call stacks B/C/E):

  prefill : RadixCache.match_prefix -> inc_lock_ref -> alloc_for_extend
            (req slots, KV slots with eviction, write_req_to_token) ->
            ForwardBatch.init_new -> model.forward -> Sampler ->
            RadixCache.cache_unfinished_req
  decode  : alloc_for_decode -> (hipGraph replay | eager forward) -> Sampler
  finish  : RadixCache.cache_finished_req -> ReqToTokenPool.free

Reference files: srt/managers/scheduler.py:3225-3711 (batching), schedule_batch.py:2378,
3060 (prepare_for_extend / prepare_for_decode), mem_cache/allocation.py:281-381,512-560,
model_executor/model_runner.py:1520-1792.  The real scheduler (zmq, tokenizer,
policies) is out of scope; this file is the harness that feeds the same data
structures with the same call order.
"""
from __future__ import annotations

import time
from array import array
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import torch

from .. import kernels
from ..distributed import parallel_state as ps
from ..layers.attention.hip_backend import HipAttnBackend
from ..layers.sampler import Sampler, SamplingBatchInfo, create_sampler
from ..mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator
from ..mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool
from ..mem_cache.radix_cache import EvictParams, MatchPrefixParams, RadixCache, RadixKey
from ..model_executor.forward_batch_info import ForwardBatch, ForwardMode, VerifyInput
from ..model_executor.graph_runner import DecodeGraphRunner
from .models import CONFIGS, CausalLM, ModelConfig


@dataclass
class Req:
    """The fields of srt/managers/schedule_batch.py Req that the cache / pools touch."""

    rid: int
    origin_input_ids: List[int]
    max_new_tokens: int
    output_ids: List[int] = field(default_factory=list)
    req_pool_idx: Optional[int] = None
    prefix_indices: Optional[torch.Tensor] = None
    last_node: Any = None
    cache_protected_len: int = 0
    extra_key: Optional[str] = None
    cache_salt: Optional[str] = None
    priority: int = 0
    # timing (TTFT)
    t_arrive: float = 0.0
    t_first_token: float = 0.0
    cached_tokens: int = 0
    fill_len: int = 0            # chunked prefill: tokens of the prompt committed so far (0 = not truncated)
    retracted_output_ids: List[int] = field(default_factory=list)   # outputs folded into the prompt by a retraction
    mm_items: Optional[List[Any]] = None   # images of the request (harness/llava.py MultimodalItem); prompt ids hold their pad values

    @property
    def all_output_ids(self) -> List[int]:
        """Every token generated for this request, across retractions."""
        return self.retracted_output_ids + self.output_ids

    @property
    def origin_array(self):
        """The prompt as a C array of int64, built once: radix keys are slices of it (a memcpy) instead of a
        fresh list -> array conversion per match / insert."""
        arr = self.__dict__.get("_origin_array")
        if arr is None or len(arr) != len(self.origin_input_ids):
            arr = self.__dict__["_origin_array"] = array("q", self.origin_input_ids)
        return arr

    def get_fill_ids(self):
        """schedule_batch.py Req.fill_ids = origin_input_ids + output_ids (here as an int64 array)."""
        arr = self.origin_array + array("q", self.output_ids) if self.output_ids else self.origin_array
        return arr[: self.fill_len] if self.fill_len else arr

    @property
    def seqlen(self) -> int:
        return len(self.origin_input_ids) + len(self.output_ids)

    def finished(self) -> bool:
        return len(self.output_ids) >= self.max_new_tokens


class ModelRunner:
    def __init__(self, config: ModelConfig, *, max_total_tokens: int, max_running_requests: int,
                 max_context_len: int, page_size: int = 1, device=None, init_device=None,
                 use_graph: bool = True, graph_max_bs: Optional[int] = None, disable_radix_cache: bool = False,
                 strict_graph: bool = False, kv_cache_dtype: str = "auto", use_hnd: bool = False, vision_config=None):
        self.config = config
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type == "cuda":
            from ..tuning import load_gemm_selections

            load_gemm_selections()        # library-GEMM selections for the prefill-sized projections (lookup only)
        self.tp_rank = ps.get_tensor_model_parallel_rank()
        self.tp_size = ps.get_tensor_model_parallel_world_size()
        self.page_size = page_size
        self.model = CausalLM(config, self.device, init_device, self.tp_rank, self.tp_size)
        self.vision = None
        if vision_config is not None:            # LLaVA: CLIP tower + projector in front of the language model
            from .llava import LlavaVision

            self.vision = LlavaVision(vision_config, config.hidden_size, self.device, init_device)
        self.num_attention_heads_per_rank = self.model.num_attention_heads_per_rank
        self.num_kv_heads_per_rank = self.model.num_kv_heads_per_rank
        self.head_dim = config.head_dim
        # pools (model_runner.py:822 alloc_memory_pool)
        self.req_to_token_pool = ReqToTokenPool(max_running_requests, max_context_len, self.device)
        size = max_total_tokens // page_size * page_size
        kv_dtype = {"auto": torch.bfloat16, "bfloat16": torch.bfloat16, "fp8_e4m3": torch.float8_e4m3fn}[kv_cache_dtype]
        self.token_to_kv_pool = MHATokenToKVPool(size, page_size, kv_dtype, self.num_kv_heads_per_rank,
                                                 config.head_dim, config.num_hidden_layers, self.device, use_hnd=use_hnd)
        if page_size == 1:
            self.token_to_kv_pool_allocator = TokenToKVPoolAllocator(size, torch.bfloat16, self.device,
                                                                     self.token_to_kv_pool)
        else:
            self.token_to_kv_pool_allocator = PagedTokenToKVPoolAllocator(size, page_size, torch.bfloat16,
                                                                          self.device, self.token_to_kv_pool)
        self.tree_cache = RadixCache(self.req_to_token_pool, self.token_to_kv_pool_allocator, page_size,
                                     disable=disable_radix_cache)
        self.attn_backend = HipAttnBackend(self)                      # model_runner.py:942
        self.sampler: Sampler = create_sampler("hip")                 # model_runner.py:651
        self.graph_runner = None
        if use_graph:
            try:
                self.graph_runner = DecodeGraphRunner(self, graph_max_bs or max_running_requests)  # :1007
            except RuntimeError as e:
                if strict_graph:             # a benchmark must not silently measure the eager path
                    raise
                # e.g. a collective that cannot be captured on this RCCL build: decode eagerly instead of dying
                # (model_runner.py falls back the same way when cuda graph capture fails)
                import warnings

                warnings.warn(f"hipGraph capture failed, decoding eagerly: {e}")
                torch.cuda.synchronize()
                self.graph_runner = None

    # ------------------------------------------------------------------ forward (model_runner.py:1520-1790)
    def forward(self, fb: ForwardBatch):
        if fb.forward_mode.is_decode() and self.graph_runner is not None and self.graph_runner.can_run(fb.batch_size):
            return self.graph_runner.replay(fb)
        self.attn_backend.init_forward_metadata(fb)
        return self.model.forward(fb.input_ids, fb.positions, fb)

    def sample(self, logits_output, fb: ForwardBatch) -> torch.Tensor:
        info = fb.sampling_info
        if info is None:
            # the default (all-greedy) batch info is four constant tensors: built once per batch size, not once per
            # decode step (four fill launches = 18 us of every 4.3 ms step)
            cache = self.__dict__.setdefault("_greedy_info", {})
            info = cache.get(fb.batch_size)
            if info is None:
                info = cache[fb.batch_size] = SamplingBatchInfo.greedy(fb.batch_size, self.device)
        pos = fb.positions if fb.forward_mode.is_decode() else None
        if pos is None and info.sampling_seed is not None:
            pos = (fb.seq_lens - 1).to(torch.int64)
        return self.sampler(logits_output, info, positions=pos)


class Engine:
    """Runs batches of requests to completion (prefill then decode), like one TP
    rank's scheduler process.  All ranks run it in lock-step with identical inputs."""

    def __init__(self, runner: ModelRunner):
        self.r = runner
        self.device = runner.device
        self.running: List[Req] = []
        self.waiting: List[Req] = []          # retracted requests waiting to be prefilled again
        self.stats: Dict[str, float] = {}
        self.logits_trace: Optional[List[torch.Tensor]] = None   # tests: set to [] to record every step's logits
        self.logits_device_trace: Optional[List[torch.Tensor]] = None   # same, kept on the device in the model dtype
        self.logits_by_req: Optional[Dict[int, List[torch.Tensor]]] = None  # tests: {} -> per request, one row per sampled token
        self._deferred_prefill: List[tuple] = []   # (requests, pinned ids, event) of prefill(defer_ids=True) passes
        self._pin_ids: Optional[torch.Tensor] = None
        self._pin_used = 0

    def _mm_embeds(self, reqs, input_ids, prefix_lens, extend_lens):
        """Image + text batches: the extend tokens' embeddings with the image features scattered over the pad-value
        positions (mm_utils.py:463-503); None for text-only batches."""
        if getattr(self.r, "vision", None) is None or not any(getattr(q, "mm_items", None) for q in reqs):
            return None
        from .llava import embed_mm_inputs

        return embed_mm_inputs(input_ids, self.r.model.embed_tokens.data, [getattr(q, "mm_items", None) for q in reqs],
                               prefix_lens, extend_lens, self.r.vision)

    def _record_logits(self, logits, reqs=None, sampled=None) -> None:
        if self.logits_trace is not None:
            self.logits_trace.append(logits.next_token_logits.float().cpu())
        if self.logits_device_trace is not None:
            self.logits_device_trace.append(logits.next_token_logits.clone())   # the graph's output buffer is reused
        if self.logits_by_req is not None and reqs is not None:
            rows = logits.next_token_logits.float().cpu()
            for i, q in enumerate(reqs):
                if sampled is None or sampled[i]:            # a truncated chunk samples nothing
                    self.logits_by_req.setdefault(q.rid, []).append(rows[i])

    # ---- KV slot allocation with eviction (allocation.py:150-279 alloc_token_slots) ----
    @staticmethod
    def _extend_ids(reqs: Sequence[Req], prefix_lens: Sequence[int]) -> torch.Tensor:
        """int64 [sum of extend lengths]: every request's prompt behind its cached prefix, concatenated (host tensor).  Buffer copies of
        the requests' `array('q')` prompts instead of a Python loop over the tokens (0.4 ms at 7680 tokens)."""
        import numpy as np

        parts = []
        for q, p in zip(reqs, prefix_lens):
            arr = getattr(q, "origin_array", None)
            if arr is not None and len(arr) == len(q.origin_input_ids):
                parts.append(np.frombuffer(arr, dtype=np.int64)[p:])
            else:
                parts.append(np.asarray(q.origin_input_ids[p:], dtype=np.int64))
        if not parts:
            return torch.empty(0, dtype=torch.int64)
        return torch.from_numpy(np.concatenate(parts))

    def _alloc_token_slots(self, n: int) -> torch.Tensor:
        alloc, tree = self.r.token_to_kv_pool_allocator, self.r.tree_cache
        if alloc.available_size() < n:
            tree.evict(EvictParams(num_tokens=n - alloc.available_size()))
        out = alloc.alloc(n)
        if out is None:
            raise RuntimeError(f"out of KV slots: need {n}, available {alloc.available_size()}, "
                               f"evictable {tree.evictable_size()}")
        return out

    # ---- prefill (scheduler.py:3225 get_new_batch_prefill + schedule_batch.py:2378) ----
    def prefill(self, reqs: Sequence[Req], sampling_info: Optional[SamplingBatchInfo] = None,
                defer_ids: bool = False) -> torch.Tensor:
        """One extend pass over whole prompts.  With `defer_ids` the sampled ids are handed to the requests LATER -- by
        the next pass (once its own forward is queued), the next decode step or `resolve_prefill_ids()` -- so the host
        prepares and launches the next pass while the GPU is still inside this one (the reference's overlap loop:
        scheduler.py event_loop_overlap processes batch N's result after launching batch N + 1)."""
        r, dev, ps_ = self.r, self.device, self.r.page_size
        tree = r.tree_cache
        self._retire_decode_state(keep_deferred=True)
        for q in reqs:
            # match against at most len-1 tokens so at least one token is computed (schedule_policy.py:138)
            key = RadixKey(q.origin_array[: len(q.origin_input_ids) - 1], q.extra_key, q.cache_salt)
            m = tree.match_prefix(MatchPrefixParams(key=key))
            q.prefix_indices, q.last_node = m.device_indices, m.last_device_node
            q.cached_tokens = int(m.device_indices.numel())
            q.cache_protected_len = q.cached_tokens
            tree.inc_lock_ref(q.last_node)
        if r.req_to_token_pool.alloc(reqs) is None:
            raise RuntimeError("out of request slots")
        prefix_lens = [int(q.prefix_indices.numel()) for q in reqs]
        seq_lens = [len(q.origin_input_ids) for q in reqs]
        extend_lens = [s - p for s, p in zip(seq_lens, prefix_lens)]
        T = sum(extend_lens)
        req_pool_cpu = torch.tensor([q.req_pool_idx for q in reqs], dtype=torch.int64)
        prefix_cpu = torch.tensor(prefix_lens, dtype=torch.int64)
        seq_cpu = torch.tensor(seq_lens, dtype=torch.int64)
        ext_cpu = torch.tensor(extend_lens, dtype=torch.int64)
        req_pool_dev = req_pool_cpu.to(dev, non_blocking=True)
        prefix_dev, seq_dev, ext_dev = (t.to(dev, non_blocking=True) for t in (prefix_cpu, seq_cpu, ext_cpu))
        # alloc_for_extend (allocation.py:281-381)
        if ps_ == 1:
            out_cache_loc = self._alloc_token_slots(T)
        else:
            alloc = r.token_to_kv_pool_allocator
            need_pages = sum((s + ps_ - 1) // ps_ - (p + ps_ - 1) // ps_ for s, p in zip(seq_lens, prefix_lens))
            if len(alloc.free_pages) < need_pages:
                tree.evict(EvictParams(num_tokens=(need_pages - len(alloc.free_pages)) * ps_))
            last_loc = torch.cat([(q.prefix_indices[-1:] if q.prefix_indices.numel() > 0
                                   else torch.full((1,), -1, dtype=torch.int64, device=dev)) for q in reqs])
            out_cache_loc = alloc.alloc_extend(prefix_dev, prefix_cpu, seq_dev, seq_cpu, last_loc, T, need_pages)
            if out_cache_loc is None:
                raise RuntimeError("out of KV pages")
        # write_cache_indices (allocation.py:54-103) as one gfx950 kernel
        prefix_ptrs = torch.tensor([q.prefix_indices.data_ptr() if q.prefix_indices.numel() else 0 for q in reqs],
                                   dtype=torch.int64).to(dev, non_blocking=True)
        kernels.write_req_to_token(r.req_to_token_pool.req_to_token, req_pool_dev, prefix_ptrs, prefix_dev, seq_dev,
                                   ext_dev, out_cache_loc)
        input_ids = self._extend_ids(reqs, prefix_lens).to(dev, non_blocking=True)
        fb = ForwardBatch.init_new(forward_mode=ForwardMode.EXTEND, input_ids=input_ids, req_pool_indices=req_pool_dev,
                                   seq_lens=seq_dev.to(torch.int32), out_cache_loc=out_cache_loc, seq_lens_cpu=seq_cpu,
                                   req_to_token_pool=r.req_to_token_pool, token_to_kv_pool=r.token_to_kv_pool,
                                   attn_backend=r.attn_backend, extend_prefix_lens_cpu=prefix_lens,
                                   extend_seq_lens_cpu=extend_lens, sampling_info=sampling_info)
        fb.input_embeds = self._mm_embeds(reqs, input_ids, prefix_lens, extend_lens)
        logits = r.forward(fb)
        self._record_logits(logits, reqs)
        next_ids = r.sample(logits, fb)
        # process_batch_result_prefill -> cache_unfinished_req (batch_result_processor.py:240).  The insert takes the PROMPT only (the new
        # token has no KV yet), so it does not wait for the sampled ids: it runs on the host while the GPU is still inside the forward
        # launched above (its few device ops queue up behind it) -- the way the reference's overlap loop processes a batch's
        # bookkeeping under the next launch.  Round 6: 1.6 ms of a 89 ms warm pass at 60 requests used to sit behind the sync.
        for q in reqs:
            saved = q.output_ids
            q.output_ids = []                        # fill_ids = prompt only
            tree.cache_unfinished_req(q)
            q.output_ids = saved
        if defer_ids:
            host = self._pinned_ids(len(reqs))
            host.copy_(next_ids, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._deferred_prefill.append((list(reqs), host, ev))
            self.running.extend(reqs)
            return next_ids
        self.resolve_prefill_ids()                   # an earlier pass's ids: its forward finished long ago
        ids_cpu = next_ids.tolist()                  # the scheduler's one sync per step
        now = time.perf_counter()
        for q, t in zip(reqs, ids_cpu):
            q.output_ids.append(int(t))
            q.t_first_token = now
        self.running.extend(reqs)
        return next_ids

    def _pinned_ids(self, n: int) -> torch.Tensor:
        """A slice of one page-locked id buffer (a fresh hipHostMalloc per pass costs more than the pass's bookkeeping)."""
        cap = max(1024, self.r.req_to_token_pool.req_to_token.shape[0])
        if self._pin_ids is None or self._pin_used + n > self._pin_ids.numel():
            if self._deferred_prefill and self._pin_ids is not None:
                self.resolve_prefill_ids()
            if self._pin_ids is None or n > self._pin_ids.numel():
                self._pin_ids = torch.empty(max(cap, n), dtype=torch.int64, pin_memory=True)
            self._pin_used = 0
        out = self._pin_ids[self._pin_used:self._pin_used + n]
        self._pin_used += n
        return out

    def resolve_prefill_ids(self) -> None:
        """Deliver the ids of every prefill pass run with defer_ids=True (waits for those passes only)."""
        if not self._deferred_prefill:
            return
        for reqs, host, ev in self._deferred_prefill:
            ev.synchronize()
            now = time.perf_counter()
            for q, t in zip(reqs, host.tolist()):
                q.output_ids.append(int(t))
                q.t_first_token = now
        self._deferred_prefill = []
        self._pin_used = 0

    # ---- chunked prefill (schedule_policy.py:1004-1060 add_chunked_req, :1160-1200 add_one_req) ----
    def prefill_chunked(self, reqs: Sequence[Req], chunked_prefill_size: int,
                        sampling_info: Optional[SamplingBatchInfo] = None) -> None:
        """Prefill with a per-pass token budget: every pass extends whole requests while the budget lasts; the
        request that does not fit is truncated to what is left (page aligned) and continues FIRST in the next
        pass (at most one chunked request at a time, like the reference's PrefillAdder).  A truncated request's
        committed tokens go into the radix tree with cache_unfinished_req(chunked=True); only requests whose
        prompt completes in a pass are sampled and join the running batch."""
        assert chunked_prefill_size >= self.r.page_size
        pending = list(reqs)
        chunked: Optional[Req] = None
        while pending or chunked is not None:
            budget = chunked_prefill_size
            batch: List[Req] = []
            ends: List[int] = []
            if chunked is not None:
                q = chunked
                chunked = None
                left = len(q.origin_input_ids) - len(q.prefix_indices)
                take = min(left, budget)
                if take < left:
                    take = take // self.r.page_size * self.r.page_size
                    chunked = q
                batch.append(q)
                ends.append(len(q.prefix_indices) + take)
                budget -= take
            while pending and budget > 0 and chunked is None:
                q = pending.pop(0)
                self._match_and_lock(q)
                left = len(q.origin_input_ids) - len(q.prefix_indices)
                take = min(left, budget)
                if take < left:
                    take = take // self.r.page_size * self.r.page_size
                    chunked = q
                    if take == 0:                      # not even a page of budget left: it starts the next pass
                        break
                batch.append(q)
                ends.append(len(q.prefix_indices) + take)
                budget -= take
            if batch:
                self._extend_pass(batch, ends, sampling_info)

    def _match_and_lock(self, q: Req) -> None:
        key = RadixKey(q.origin_array[: len(q.origin_input_ids) - 1], q.extra_key, q.cache_salt)
        m = self.r.tree_cache.match_prefix(MatchPrefixParams(key=key))
        q.prefix_indices, q.last_node = m.device_indices, m.last_device_node
        q.cached_tokens = int(m.device_indices.numel())
        q.cache_protected_len = q.cached_tokens
        self.r.tree_cache.inc_lock_ref(q.last_node)

    def _extend_pass(self, reqs: Sequence[Req], ends: Sequence[int], sampling_info) -> None:
        """One extend forward over `reqs`, request i committing prompt tokens [len(prefix_indices), ends[i])."""
        r, dev, ps_ = self.r, self.device, self.r.page_size
        tree = r.tree_cache
        self._retire_decode_state()
        fresh = [q for q in reqs if q.req_pool_idx is None]
        if fresh and r.req_to_token_pool.alloc(fresh) is None:
            raise RuntimeError("out of request slots")
        prefix_lens = [int(q.prefix_indices.numel()) for q in reqs]
        seq_lens = list(ends)
        extend_lens = [s - p for s, p in zip(seq_lens, prefix_lens)]
        T = sum(extend_lens)
        req_pool_cpu = torch.tensor([q.req_pool_idx for q in reqs], dtype=torch.int64)
        prefix_cpu = torch.tensor(prefix_lens, dtype=torch.int64)
        seq_cpu = torch.tensor(seq_lens, dtype=torch.int64)
        ext_cpu = torch.tensor(extend_lens, dtype=torch.int64)
        req_pool_dev = req_pool_cpu.to(dev, non_blocking=True)
        prefix_dev, seq_dev, ext_dev = (t.to(dev, non_blocking=True) for t in (prefix_cpu, seq_cpu, ext_cpu))
        if ps_ == 1:
            out_cache_loc = self._alloc_token_slots(T)
        else:
            alloc = r.token_to_kv_pool_allocator
            need_pages = sum((s + ps_ - 1) // ps_ - (p + ps_ - 1) // ps_ for s, p in zip(seq_lens, prefix_lens))
            if len(alloc.free_pages) < need_pages:
                tree.evict(EvictParams(num_tokens=(need_pages - len(alloc.free_pages)) * ps_))
            last_loc = torch.cat([(q.prefix_indices[-1:] if q.prefix_indices.numel() > 0
                                   else torch.full((1,), -1, dtype=torch.int64, device=dev)) for q in reqs])
            out_cache_loc = alloc.alloc_extend(prefix_dev, prefix_cpu, seq_dev, seq_cpu, last_loc, T, need_pages)
            if out_cache_loc is None:
                raise RuntimeError("out of KV pages")
        prefix_ptrs = torch.tensor([q.prefix_indices.data_ptr() if q.prefix_indices.numel() else 0 for q in reqs],
                                   dtype=torch.int64).to(dev, non_blocking=True)
        kernels.write_req_to_token(r.req_to_token_pool.req_to_token, req_pool_dev, prefix_ptrs, prefix_dev, seq_dev,
                                   ext_dev, out_cache_loc)
        input_ids = torch.tensor([t for q, p, e in zip(reqs, prefix_lens, seq_lens) for t in q.origin_input_ids[p:e]],
                                 dtype=torch.int64).to(dev, non_blocking=True)
        fb = ForwardBatch.init_new(forward_mode=ForwardMode.EXTEND, input_ids=input_ids, req_pool_indices=req_pool_dev,
                                   seq_lens=seq_dev.to(torch.int32), out_cache_loc=out_cache_loc, seq_lens_cpu=seq_cpu,
                                   req_to_token_pool=r.req_to_token_pool, token_to_kv_pool=r.token_to_kv_pool,
                                   attn_backend=r.attn_backend, extend_prefix_lens_cpu=prefix_lens,
                                   extend_seq_lens_cpu=extend_lens, sampling_info=sampling_info)
        fb.input_embeds = self._mm_embeds(reqs, input_ids, prefix_lens, extend_lens)
        logits = r.forward(fb)
        self._record_logits(logits, reqs, [end >= len(q.origin_input_ids) for q, end in zip(reqs, ends)])
        ids_cpu = r.sample(logits, fb).tolist()
        now = time.perf_counter()
        for q, end, t in zip(reqs, ends, ids_cpu):
            if end < len(q.origin_input_ids):          # truncated: commit the chunk, no token yet
                q.fill_len = end
                tree.cache_unfinished_req(q, chunked=True)
                continue
            q.fill_len = 0
            q.t_first_token = now
            tree.cache_unfinished_req(q)               # fill_ids = prompt only: the new token has no KV yet
            q.output_ids.append(int(t))
            self.running.append(q)

    # ---- decode (scheduler.py:3566 update_running_batch + schedule_batch.py:3060) ----
    def decode_step(self, sampling_info: Optional[SamplingBatchInfo] = None) -> torch.Tensor:
        r, dev, ps_ = self.r, self.device, self.r.page_size
        self.resolve_prefill_ids()
        if not self.check_decode_mem():
            before = list(self.running)
            self.waiting.extend(self.retract_decode())
            if sampling_info is not None:               # the per-batch sampling parameters follow the survivors
                pos = {id(q): i for i, q in enumerate(before)}
                sampling_info = sampling_info.filter_batch([pos[id(q)] for q in self.running])
        reqs = self.running
        bs = len(reqs)
        st = self._decode_state
        if st is None or st["bs"] != bs:
            # hand-offs still in flight belong to the OLD batch composition: deliver them before the state
            # (and its pending list) is rebuilt, or those tokens never reach output_ids
            self.flush_decode_outputs()
            st = self._decode_state = self._build_decode_state(reqs)
        # alloc_for_decode (allocation.py:512-560): one slot per request, written at column seq_len
        fused = st.get("fused", False)
        if ps_ == 1 and fused:
            out_cache_loc = st["out_cache_loc"]
            kernels.decode_advance(r.req_to_token_pool.req_to_token, st["req_pool"], st["seq_lens"],
                                   self._alloc_token_slots(bs), out_cache_loc)
        elif ps_ == 1:
            out_cache_loc = self._alloc_token_slots(bs)
        else:
            last_loc = r.req_to_token_pool.req_to_token[st["req_pool"], (st["seq_lens"] - 1).long()].to(torch.int64)
            alloc = r.token_to_kv_pool_allocator
            if len(alloc.free_pages) < bs:
                r.tree_cache.evict(EvictParams(num_tokens=bs * ps_))
            out_cache_loc = alloc.alloc_decode((st["seq_lens"] + 1).to(torch.int64), st["seq_lens_cpu"] + 1, last_loc)
            if out_cache_loc is None:
                raise RuntimeError("out of KV pages")
        if not fused:
            r.req_to_token_pool.req_to_token[st["req_pool"], st["seq_lens"].long()] = out_cache_loc.to(torch.int32)
            st["seq_lens"] += 1
        st["seq_lens_cpu"] += 1
        fb = ForwardBatch(forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=st["last_ids"],
                          req_pool_indices=st["req_pool"], seq_lens=st["seq_lens"], out_cache_loc=out_cache_loc,
                          seq_lens_sum=int(st["seq_lens_cpu"].sum()), seq_lens_cpu=st["seq_lens_cpu"],
                          req_to_token_pool=r.req_to_token_pool, token_to_kv_pool=r.token_to_kv_pool,
                          attn_backend=r.attn_backend, sampling_info=sampling_info)
        if r.graph_runner is None or not r.graph_runner.can_run(bs):
            fb.positions = kernels.clamp_position(fb.seq_lens)
        replayed = r.graph_runner is not None and r.graph_runner.can_run(bs)
        logits = r.forward(fb)
        self._record_logits(logits, st["reqs"])
        info = sampling_info
        if (fused and replayed and getattr(r.graph_runner, "greedy_ids_in_graph", False) and (info is None or info.is_all_greedy)
                and st["last_ids"].data_ptr() == r.graph_runner.input_ids.data_ptr()):
            # the replayed graph has already put the greedy pick of every row into its own input buffer (= this state's last_ids):
            # nothing to launch, nothing to copy.  (The returned tensor IS that buffer: valid until the next step.)
            next_ids = st["last_ids"]
        else:
            next_ids = r.sample(logits, fb)
            if fused:
                st["last_ids"].copy_(next_ids)          # straight into the graph's input buffer
            else:
                st["last_ids"] = next_ids.to(torch.int64)
        # token hand-off to the host: an async copy into pinned memory + an event, so that the scheduler can
        # launch step N+1 before it looks at step N's tokens (overlap scheduling, scheduler.py:1783)
        on_gpu = next_ids.is_cuda                       # (host-logic tests drive this class with CPU tensors)
        host = st["free_host"].pop() if st["free_host"] else torch.empty(next_ids.shape, dtype=next_ids.dtype,
                                                                          pin_memory=on_gpu)
        host.copy_(next_ids, non_blocking=True)
        ev = None
        if on_gpu:
            ev = torch.cuda.Event()
            ev.record()
        st["pending"].append((host, ev))
        return next_ids

    def _build_decode_state(self, reqs: Sequence[Req]):
        dev = self.device
        bs = len(reqs)
        seq = [q.seqlen - 1 for q in reqs]   # KV rows present: every token except the newest sampled one
        st = dict(bs=bs, req_pool=torch.tensor([q.req_pool_idx for q in reqs], dtype=torch.int64, device=dev),
                  seq_lens=torch.tensor(seq, dtype=torch.int32, device=dev),
                  seq_lens_cpu=torch.tensor(seq, dtype=torch.int64),
                  last_ids=torch.tensor([q.output_ids[-1] for q in reqs], dtype=torch.int64, device=dev),
                  pending=[], free_host=[], reqs=list(reqs), fused=False)
        # On the GPU the per-batch device state LIVES in the decode graph's static input buffers (the replay then has
        # nothing to copy) and one launch does the step's slot bookkeeping (kernels.decode_advance): the loop between
        # two replays was a dozen eager launches of ~4.5 us each.
        gr = getattr(self.r, "graph_runner", None)
        import os

        if torch.device(dev).type == "cuda" and self.r.page_size == 1 and os.environ.get("SGLANG_AMD_DECODE_FUSED_PREP", "1") != "0":
            if gr is not None and gr.can_run(bs):
                for key, buf in (("req_pool", gr.req_pool_indices), ("seq_lens", gr.seq_lens), ("last_ids", gr.input_ids)):
                    view = buf[:bs]
                    view.copy_(st[key])
                    st[key] = view
                st["out_cache_loc"] = gr.out_cache_loc[:bs]
            else:
                st["out_cache_loc"] = torch.zeros(bs, dtype=torch.int64, device=dev)
            st["fused"] = True
        return st

    _decode_state = None

    def _retire_decode_state(self, keep_deferred: bool = False) -> None:
        """The running batch is about to change (new requests join): deliver every in-flight hand-off to the
        requests it was sampled for, then drop the per-batch device state so that the next decode step rebuilds
        it from the requests' own output_ids."""
        if not keep_deferred:
            self.resolve_prefill_ids()
        if self._decode_state is not None:
            self.flush_decode_outputs()
            self._decode_state = None

    def flush_decode_outputs(self, lag: int = 0) -> None:
        """Hand the sampled ids of the finished steps to the requests.  `lag` = newest steps left in flight:
        with lag=1 the host only waits for the step BEFORE the one it has just launched, so the GPU never idles
        on the host's per-step bookkeeping (the reference's overlap scheduler)."""
        st = self._decode_state
        if not st or len(st["pending"]) <= lag:
            return
        n = len(st["pending"]) - lag
        ready, st["pending"] = st["pending"][:n], st["pending"][n:]
        for host, ev in ready:
            if ev is not None:
                ev.synchronize()
            for q, t in zip(st["reqs"], host.tolist()):
                q.output_ids.append(t)
            st["free_host"].append(host)

    # ---- retraction under pool pressure (schedule_batch.py:2825-2925 retract_decode, mem_cache/common.py:198) ----
    def _kv_lens(self) -> List[int]:
        """Tokens with a KV row per running request.  With hand-offs in flight (flush_decode_outputs(lag >= 1)) the
        requests' output_ids trail the device by the pending steps: the decode state's host-side lengths are exact."""
        st = self._decode_state
        if st is not None and st["bs"] == len(self.running) and all(a is b for a, b in zip(st["reqs"], self.running)):
            return [int(x) for x in st["seq_lens_cpu"].tolist()]
        return [q.seqlen - 1 for q in self.running]

    def new_tokens_required_next_decode(self) -> int:
        """schedule_batch.py:2791-2803: one slot per request, a whole page when its last page is full."""
        ps_ = self.r.page_size
        return sum(1 for n in self._kv_lens() if n % ps_ == 0) * ps_

    def check_decode_mem(self) -> bool:
        alloc, tree = self.r.token_to_kv_pool_allocator, self.r.tree_cache
        return alloc.available_size() + tree.evictable_size() >= self.new_tokens_required_next_decode()

    def retract_decode(self) -> List[Req]:
        """Not enough KV slots for the next decode step even after evicting the whole tree: give up the requests that
        are cheapest to redo -- fewest generated tokens first, longer prompts first among equals (the reference's
        (len(output_ids), -len(origin_input_ids)) order, popped from the end) -- until the rest fits; always keep one.
        A retracted request's KV is freed without inserting it into the tree ("we need the space instantly"), its
        generated tokens become part of its prompt, and it waits in `self.waiting` to be prefilled again."""
        self.flush_decode_outputs()
        order = sorted(range(len(self.running)), key=lambda i: (len(self.running[i].output_ids),
                                                                 -len(self.running[i].origin_input_ids)), reverse=True)
        tree, pool, alloc = self.r.tree_cache, self.r.req_to_token_pool, self.r.token_to_kv_pool_allocator
        retracted: List[Req] = []
        first = True
        while first or not self._fits([self.running[i] for i in order]):
            if len(order) == 1:
                break
            first = False
            q = self.running[order.pop()]
            tree.cache_finished_req(q, is_insert=False, kv_len_to_handle=q.seqlen - 1)
            pool.free(q)
            q.retracted_output_ids += q.output_ids
            q.origin_input_ids = list(q.origin_input_ids) + q.output_ids
            q.max_new_tokens -= len(q.output_ids)
            q.output_ids = []
            q.prefix_indices, q.last_node, q.cache_protected_len, q.fill_len = None, None, 0, 0
            retracted.append(q)
        keep = sorted(order)
        self.running = [self.running[i] for i in keep]
        self._decode_state = None
        self.stats["retracted"] = self.stats.get("retracted", 0) + len(retracted)
        return retracted

    def _fits_with(self, q: Req) -> bool:
        """Room to prefill a waiting request AND run the next decode step of everyone (PrefillAdder's budget check)."""
        alloc, tree = self.r.token_to_kv_pool_allocator, self.r.tree_cache
        need = len(q.origin_input_ids) + self.new_tokens_required_next_decode() + len(self.running) + 1
        return alloc.available_size() + tree.evictable_size() >= need and self.r.req_to_token_pool.available_size() > 0

    def _fits(self, reqs: Sequence[Req]) -> bool:
        alloc, tree, ps_ = self.r.token_to_kv_pool_allocator, self.r.tree_cache, self.r.page_size
        need = sum(1 for q in reqs if (q.seqlen - 1) % ps_ == 0) * ps_
        return alloc.available_size() + tree.evictable_size() >= need

    # ---- mixed batch (schedule_batch.py:2758-2789 mix_with_running, ForwardMode.MIXED) ----
    def mixed_step(self, new_reqs: Sequence[Req], sampling_info: Optional[SamplingBatchInfo] = None) -> None:
        """One forward that prefills `new_reqs` AND advances every running request by one token: the running ones join
        the extend batch as extends of length 1 over their cached prefix (prefix_len = tokens with KV, input = their
        last sampled token, one new slot each) -- the reference's chunked-prefill "mixed" mode."""
        assert self.r.page_size == 1, "mixed batches are wired for page_size 1"
        r, dev = self.r, self.device
        tree = r.tree_cache
        self._retire_decode_state()
        run = list(self.running)
        for q in new_reqs:
            self._match_and_lock(q)
        if r.req_to_token_pool.alloc(list(new_reqs)) is None:
            raise RuntimeError("out of request slots")
        reqs = list(new_reqs) + run
        prefix_lens = [int(q.prefix_indices.numel()) for q in new_reqs] + [q.seqlen - 1 for q in run]
        seq_lens = [len(q.origin_input_ids) for q in new_reqs] + [q.seqlen for q in run]
        extend_lens = [s - p for s, p in zip(seq_lens, prefix_lens)]
        T = sum(extend_lens)
        out_cache_loc = self._alloc_token_slots(T)
        req_pool_cpu = torch.tensor([q.req_pool_idx for q in reqs], dtype=torch.int64)
        prefix_cpu, seq_cpu, ext_cpu = (torch.tensor(v, dtype=torch.int64) for v in (prefix_lens, seq_lens, extend_lens))
        req_pool_dev = req_pool_cpu.to(dev, non_blocking=True)
        prefix_dev, seq_dev, ext_dev = (t.to(dev, non_blocking=True) for t in (prefix_cpu, seq_cpu, ext_cpu))
        # new requests: prefix slots come from the tree; running ones already have their row (prefix pointer 0 = keep)
        ptrs = [q.prefix_indices.data_ptr() if q.prefix_indices.numel() else 0 for q in new_reqs] + [0] * len(run)
        kernels.write_req_to_token(r.req_to_token_pool.req_to_token, req_pool_dev,
                                   torch.tensor(ptrs, dtype=torch.int64).to(dev, non_blocking=True), prefix_dev,
                                   seq_dev, ext_dev, out_cache_loc)
        ids = [t for q, p in zip(new_reqs, prefix_lens) for t in q.origin_input_ids[p:]] + [q.output_ids[-1] for q in run]
        input_ids = torch.tensor(ids, dtype=torch.int64).to(dev, non_blocking=True)
        fb = ForwardBatch.init_new(forward_mode=ForwardMode.MIXED, input_ids=input_ids, req_pool_indices=req_pool_dev,
                                   seq_lens=seq_dev.to(torch.int32), out_cache_loc=out_cache_loc, seq_lens_cpu=seq_cpu,
                                   req_to_token_pool=r.req_to_token_pool, token_to_kv_pool=r.token_to_kv_pool,
                                   attn_backend=r.attn_backend, extend_prefix_lens_cpu=prefix_lens,
                                   extend_seq_lens_cpu=extend_lens, sampling_info=sampling_info)
        logits = r.forward(fb)
        self._record_logits(logits, reqs)
        ids_cpu = r.sample(logits, fb).tolist()
        now = time.perf_counter()
        for q, t in zip(reqs, ids_cpu):
            if q in run:
                q.output_ids.append(int(t))
                continue
            q.t_first_token = now
            tree.cache_unfinished_req(q)               # fill_ids = prompt only: the new token has no KV yet
            q.output_ids.append(int(t))
            self.running.append(q)

    # ---- finish (batch_result_processor.py:863 -> mem_cache/common.py:198 release_kv_cache) ----
    # ---- speculative decoding: one TARGET_VERIFY forward (speculative/eagle_info.py EagleVerifyInput.prepare_for_verify,
    #      triton_backend.py:860-919) ----------------------------------------------------------------------------
    def verify_tree(self, tree_tokens: Sequence[Sequence[int]], parents: Sequence[int]) -> torch.Tensor:
        """Score a tree of draft tokens for every running request in ONE forward of the target model.  Node 0 is the
        request's last sampled token (the only token without a KV row yet), node j > 0 a draft token whose parent is
        node parents[j] < j; a node attends to the request's whole context and to its ancestors (incl. itself).
        Returns the logits [B, nodes, vocab]: row j = the target model's distribution after the path to node j.
        The draft rows are scratch: their KV slots go back to the allocator (a real worker would keep the accepted path)."""
        r, dev = self.r, self.device
        assert r.page_size == 1, "the verify step is wired for page_size 1"
        self.flush_decode_outputs()
        self._retire_decode_state()
        reqs = self.running
        B, nd = len(reqs), len(parents)
        assert parents[0] == -1 and all(0 <= parents[j] < j for j in range(1, nd)) and all(len(t) == nd for t in tree_tokens)
        assert all(t[0] == q.output_ids[-1] for t, q in zip(tree_tokens, reqs)), "node 0 is the last sampled token"
        depth = [0] * nd
        anc = torch.zeros((nd, nd), dtype=torch.bool)
        for j in range(nd):
            a = j
            while a >= 0:
                anc[j, a] = True
                a = parents[a]
            depth[j] = int(anc[j].sum()) - 1
        kv = [q.seqlen - 1 for q in reqs]                           # tokens with a KV row
        slots = self._alloc_token_slots(B * nd)
        pool_idx = torch.tensor([q.req_pool_idx for q in reqs], dtype=torch.int64, device=dev)
        kv_dev = torch.tensor(kv, dtype=torch.int32, device=dev)
        cols = kv_dev.long()[:, None] + torch.arange(nd, device=dev)[None, :]
        r.req_to_token_pool.req_to_token[pool_idx[:, None], cols] = slots.view(B, nd).to(torch.int32)
        # the flat mask: request b's [nd, kv_b + nd] block, prefix visible, tree part = ancestors
        mask = torch.cat([torch.cat([torch.ones((nd, n), dtype=torch.bool), anc], dim=1).flatten() for n in kv]).to(dev)
        positions = (kv_dev.long()[:, None] + torch.tensor(depth, device=dev)[None, :]).flatten()
        fb = ForwardBatch(forward_mode=ForwardMode.TARGET_VERIFY, batch_size=B,
                          input_ids=torch.tensor([t for row in tree_tokens for t in row], dtype=torch.int64, device=dev),
                          req_pool_indices=pool_idx, seq_lens=kv_dev, out_cache_loc=slots, seq_lens_sum=sum(kv),
                          seq_lens_cpu=torch.tensor(kv, dtype=torch.int64), positions=positions,
                          req_to_token_pool=r.req_to_token_pool, token_to_kv_pool=r.token_to_kv_pool, attn_backend=r.attn_backend,
                          spec_info=VerifyInput(custom_mask=mask, draft_token_num=nd, positions=positions))
        fb.extend_seq_lens_cpu, fb.extend_prefix_lens_cpu, fb.extend_num_tokens = [nd] * B, list(kv), B * nd
        fb.extend_seq_lens = torch.full((B,), nd, dtype=torch.int32, device=dev)
        fb.extend_prefix_lens = kv_dev
        logits = r.forward(fb).next_token_logits
        r.token_to_kv_pool_allocator.free(slots)
        return logits.view(B, nd, -1)

    def finish(self, reqs: Sequence[Req]) -> None:
        self.resolve_prefill_ids()
        self.flush_decode_outputs()
        tree, pool = self.r.tree_cache, self.r.req_to_token_pool
        for q in reqs:
            # the last sampled token has no KV row: handle seqlen-1 rows
            tree.cache_finished_req(q, kv_len_to_handle=q.seqlen - 1)
            pool.free(q)
        done = set(id(q) for q in reqs)
        self.running = [q for q in self.running if id(q) not in done]
        self._decode_state = None

    # ---- convenience: run a set of requests to completion ---------------------------------
    def generate(self, reqs: Sequence[Req], sampling_info: Optional[SamplingBatchInfo] = None,
                 sync_every: int = 0) -> None:
        """Run `reqs` to completion.  The scheduler loop in small: requests that a decode step retracted under pool
        pressure wait in `self.waiting` and are prefilled again as soon as they fit (scheduler.py get_new_batch_prefill
        over the waiting queue); finished requests leave the batch; a per-batch `sampling_info` follows the batch
        composition through filter_batch (rows are looked up by request, so a re-admitted request keeps its own row)."""
        row_of = {id(q): i for i, q in enumerate(reqs)}
        if sampling_info is not None:
            foreign = [q.rid for q in list(self.waiting) + list(self.running) if id(q) not in row_of]
            if foreign:
                raise ValueError(f"Engine.generate: requests {foreign} were already queued / running before this call and have "
                                 f"no row in `sampling_info`; pass them in `reqs` (with their rows) or drain them first")

        def info_for(batch: Sequence[Req]):
            return sampling_info.filter_batch([row_of[id(q)] for q in batch]) if sampling_info is not None else None

        self.prefill(reqs, sampling_info)
        del sync_every                                  # kept for callers; every step is handed off before the next here
        steps = 0
        limit = 4 * sum(q.max_new_tokens for q in reqs) + 16
        cur_key, cur_info = None, None
        while self.running or self.waiting:
            steps += 1
            if steps > limit:
                raise RuntimeError("Engine.generate: no progress (pool too small for a single request?)")
            if self.waiting and (not self.running or self._fits_with(self.waiting[0])):
                q = self.waiting.pop(0)
                self.prefill([q], info_for([q]))
                continue
            self.flush_decode_outputs()                 # (this convenience loop looks at every step's tokens before the next)
            done = [q for q in self.running if q.finished()]
            if done:
                self.finish(done)
                continue
            key = tuple(id(q) for q in self.running)
            if key != cur_key:
                cur_key, cur_info = key, info_for(self.running)
            self.decode_step(cur_info)                  # (a retraction inside the step filters cur_info down to the survivors)
