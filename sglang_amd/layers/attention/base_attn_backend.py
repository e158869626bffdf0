"""AttentionBackend contract (reference:
/root/reference/python/sglang/srt/layers/attention/base_attn_backend.py:36-308)."""
from __future__ import annotations

from abc import ABC

import torch


class AttentionBackend(ABC):
    needs_cpu_seq_lens: bool = True
    extend_dummy_seqs_capped_by_req_pool: bool = False

    def init_forward_metadata(self, forward_batch):
        """Eager entry point = out-of-graph part + in-graph part (:65-71)."""
        self.init_forward_metadata_out_graph(forward_batch)
        self.init_forward_metadata_in_graph(forward_batch)

    def init_forward_metadata_out_graph(self, forward_batch, in_capture: bool = False):
        """Host-side / dynamic-shape metadata prep, runs outside graph capture (:73-93)."""

    def init_forward_metadata_in_graph(self, forward_batch):
        """Static-shape device ops that are recorded into the graph (:95-107)."""

    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int):
        raise NotImplementedError()

    def get_cuda_graph_seq_len_fill_value(self):
        raise NotImplementedError()

    def forward(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True, **kwargs):
        """Mode dispatch (:216-258)."""
        if forward_batch.forward_mode.is_idle():
            return q.new_empty(q.shape[0], layer.tp_q_head_num * layer.v_head_dim)
        if forward_batch.forward_mode.is_decode():
            return self.forward_decode(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)
        return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)

    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True, **kwargs):
        raise NotImplementedError()

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True, **kwargs):
        raise NotImplementedError()

    def forward_mixed(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        raise NotImplementedError()

    def support_triton(self) -> bool:
        return True
