"""Sampler with SGLang's interface (reference:
/root/reference/python/sglang/srt/layers/sampler.py:71-300, SamplingBatchInfo in
srt/sampling/sampling_batch_info.py), running the gfx950 sampling kernels."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List, Optional

import torch
from torch import nn

from .. import kernels

TOP_K_ALL = 1 << 30   # srt/sampling/sampling_params.py:40


@dataclass
class SamplingBatchInfo:
    temperatures: torch.Tensor            # [B, 1] fp32
    top_ps: torch.Tensor                  # [B] fp32
    top_ks: torch.Tensor                  # [B] int32 (TOP_K_ALL = whole vocab)
    min_ps: torch.Tensor                  # [B] fp32
    is_all_greedy: bool = True
    need_top_p_sampling: bool = False
    need_top_k_sampling: bool = False
    need_min_p_sampling: bool = False
    sampling_seed: Optional[torch.Tensor] = None   # [B] int64 -> deterministic gumbel sampling
    sync_token_ids_across_tp: bool = False

    def filter_batch(self, keep_indices) -> "SamplingBatchInfo":
        """The rows `keep_indices` of this batch, in that order (srt/sampling/sampling_batch_info.py filter_batch: the
        scheduler calls it when requests leave the running batch -- finished or retracted)."""
        idx = torch.as_tensor(keep_indices, dtype=torch.int64, device=self.top_ps.device)
        return SamplingBatchInfo(self.temperatures[idx], self.top_ps[idx], self.top_ks[idx], self.min_ps[idx], self.is_all_greedy,
                                 self.need_top_p_sampling, self.need_top_k_sampling, self.need_min_p_sampling,
                                 self.sampling_seed[idx] if self.sampling_seed is not None else None, self.sync_token_ids_across_tp)

    @classmethod
    def greedy(cls, batch: int, device) -> "SamplingBatchInfo":
        return cls(torch.ones((batch, 1), device=device), torch.ones(batch, device=device),
                   torch.full((batch,), TOP_K_ALL, dtype=torch.int32, device=device),
                   torch.zeros(batch, device=device), True)


@dataclass
class LogitsProcessorOutput:
    next_token_logits: torch.Tensor       # [B, vocab] fp32, or the model dtype (widened on use)
    hidden_states: Optional[torch.Tensor] = None
    # filled by the sampler when return_logprob (logits_processor.py LogitsProcessorOutput fields)
    next_token_logprobs: Optional[torch.Tensor] = None
    next_token_top_logprobs_val: Optional[List] = None
    next_token_top_logprobs_idx: Optional[List] = None
    next_token_token_ids_logprobs_val: Optional[List] = None
    next_token_token_ids_logprobs_idx: Optional[List] = None


# which route the non-greedy batches of this process took (tests / telemetry: tests/golden/ref_model.py reports it)
served = {"one_call_from_logits": 0, "softmax_then_sample": 0}


class Sampler(nn.Module):
    """sampler.py:71-300.  Dtypes follow the reference: greedy ids are int64 (torch.argmax, :139), sampled ids
    int32 (sampling_from_probs_torch / the torch top-k/top-p path).  Unseeded sampling draws one fresh 62-bit seed
    per row from torch's device generator inside kernels.top_k_top_p_min_p_sample (graph-safe philox), so
    `sampling_info.sampling_seed is None` means "non-deterministic", exactly like torch.multinomial in the
    reference's unseeded path."""

    def forward(self, logits_output: LogitsProcessorOutput, sampling_info: SamplingBatchInfo,
                return_logprob: bool = False, top_logprobs_nums: Optional[List[int]] = None,
                token_ids_logprobs: Optional[List[List[int]]] = None, positions: Optional[torch.Tensor] = None
                ) -> torch.Tensor:
        logits = logits_output.next_token_logits
        if logits.shape[0] == 0:
            return torch.empty((0,), dtype=torch.int64, device=logits.device)
        # custom logit processors + NaN handling of the reference instance this forward is bound to (sampler.py:126)
        pre = getattr(self, "_preprocess_logits", None)
        if pre is not None:
            logits = pre(logits, sampling_info)
        logprobs = None
        if sampling_info.is_all_greedy:
            ids = kernels.argmax(logits)                                    # sampler.py:133-141
            if return_logprob:
                logprobs = torch.log_softmax(logits.float(), dim=-1)        # :143-146
        else:
            # sampler.py:148-152, 211-260: div_ temperature, softmax in place, then sample from probs
            simple_sampling_case = not (sampling_info.need_top_p_sampling or sampling_info.need_top_k_sampling
                                        or sampling_info.need_min_p_sampling)
            probs = ids = None
            min_ps = sampling_info.min_ps if sampling_info.need_min_p_sampling else None
            if not return_logprob:
                # The logits of a decode-sized batch (bf16, or the fp32 the reference's LogitsProcessor hands over), no logprobs asked
                # for.  Filtered sampling: ONE native call, the probabilities are never written (and the logits not overwritten with
                # them: sampler.py:216 does that in place, nothing downstream reads them without return_logprob); otherwise bf16
                # logits are widened inside the softmax launches (exact; no separate pass)
                if not simple_sampling_case:
                    ids = kernels.sample_from_logits(logits, sampling_info.temperatures, sampling_info.top_ks, sampling_info.top_ps,
                                                     min_ps, sampling_info.sampling_seed, positions)
                    served["one_call_from_logits" if ids is not None else "softmax_then_sample"] += 1
                if ids is None and logits.dtype == torch.bfloat16:
                    probs = kernels.softmax_temperature_from_bf16(logits, sampling_info.temperatures)
            if ids is None:
                if probs is None:
                    if logits.dtype != torch.float32:
                        logits = logits.float()                             # exact widening of the bf16 logits
                    probs = kernels.softmax_temperature_(logits, sampling_info.temperatures)
                if simple_sampling_case:
                    ids = kernels.top_k_top_p_min_p_sample(probs, None, None, None, sampling_info.sampling_seed, positions,
                                                           filtered=False)
                else:
                    ids = kernels.top_k_top_p_min_p_sample(probs, sampling_info.top_ks, sampling_info.top_ps, min_ps,
                                                           sampling_info.sampling_seed, positions)
            if return_logprob:
                logprobs = torch.log(probs)                                 # :252-257
        if return_logprob:
            self._write_logprobs(logits_output, logprobs, ids, top_logprobs_nums, token_ids_logprobs)
        self._sync_token_ids_across_tp(ids, sampling_info)     # in place, like the reference's (returns None there)
        return ids

    def _write_logprobs(self, logits_output, logprobs, ids, top_logprobs_nums, token_ids_logprobs) -> None:
        """sampler.py:262-271: bound to a reference Sampler the instance's own output_logprob_processor does it;
        standalone, the same three outputs are produced with plain torch indexing (off the hot path)."""
        proc = getattr(self, "output_logprob_processor", None)
        if proc is not None:
            proc.compute_logprobs(logprobs, top_logprobs_nums, token_ids_logprobs, ids).write_output_to(logits_output)
            return
        rows = torch.arange(ids.shape[0], device=ids.device)
        logits_output.next_token_logprobs = logprobs[rows, ids.long()]
        if top_logprobs_nums and any(n > 0 for n in top_logprobs_nums):
            k = max(top_logprobs_nums)
            vals, idx = logprobs.topk(k, dim=-1)
            logits_output.next_token_top_logprobs_val = [v[:n].tolist() for v, n in zip(vals, top_logprobs_nums)]
            logits_output.next_token_top_logprobs_idx = [i[:n].tolist() for i, n in zip(idx, top_logprobs_nums)]
        if token_ids_logprobs and any(t is not None for t in token_ids_logprobs):
            logits_output.next_token_token_ids_logprobs_val = [
                logprobs[b, torch.tensor(t, device=logprobs.device)].tolist() if t is not None else []
                for b, t in enumerate(token_ids_logprobs)]
            logits_output.next_token_token_ids_logprobs_idx = [list(t) if t is not None else [] for t in token_ids_logprobs]

    def _sync_token_ids_across_tp(self, ids: torch.Tensor, sampling_info: SamplingBatchInfo) -> None:
        """sampler.py:497-512: MIN all-reduce of the ids (greedy and sampled alike) when requested (grammar / env)."""
        from ..distributed import parallel_state as ps

        if (getattr(sampling_info, "sync_token_ids_across_tp", False) and ps.get_tensor_model_parallel_world_size() > 1
                and ps.get_tp_group() is not None):
            import torch.distributed as dist

            dist.all_reduce(ids, op=dist.ReduceOp.MIN, group=ps.get_tp_group())


_SAMPLER_BACKENDS = {"hip": lambda: Sampler()}


def register_sampler_backend(name: str, factory) -> None:
    """sampler.py:531-542."""
    _SAMPLER_BACKENDS[name] = factory


def create_sampler(backend: str = "hip") -> Sampler:
    if backend not in _SAMPLER_BACKENDS:
        raise ValueError(f"Unknown sampling backend '{backend}'. Register it via register_sampler_backend().")
    return _SAMPLER_BACKENDS[backend]()
