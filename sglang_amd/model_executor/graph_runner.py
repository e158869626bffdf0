"""Decode hipGraph runner: one captured graph per padded batch-size bucket, static
input buffers, replay = copy-in + graph launch.

Mirrors /root/reference/python/sglang/srt/model_executor/runner/
decode_cuda_graph_runner.py (:200 class, :1118 capture_one_shape, :1240 load_batch,
:1386 execute), base_cuda_graph_runner.py:64-102 (bucket list) and
runner_backend/full_cuda_graph_backend.py:49-157.  torch.cuda.CUDAGraph on ROCm
is hipGraph (hipStreamBeginCapture / hipGraphLaunch); the gfx950 kernels are
launched on torch's current stream so they are recorded like any other node.

What is different here: the attention backend needs no out-of-graph metadata
refresh (it reads seq_lens / req_to_token on the device), and positions are
recomputed inside the graph from the static seq_lens buffer, so a replay is
exactly five small device copies + one hipGraphLaunch.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import kernels
from ..layers.sampler import LogitsProcessorOutput
from .forward_batch_info import ForwardBatch, ForwardMode


def get_batch_sizes_to_capture(max_bs: int) -> List[int]:
    """base_cuda_graph_runner.py:64-102 style buckets: dense small sizes, then steps of 8/16/32."""
    sizes = [1, 2, 4] + list(range(8, 33, 8)) + list(range(48, 129, 16)) + list(range(160, 513, 32))
    sizes = sorted(set(s for s in sizes if s <= max_bs) | {max_bs})
    return sizes


class DecodeGraphRunner:
    def __init__(self, model_runner, max_bs: int, capture_bs: Optional[List[int]] = None):
        self.mr = model_runner
        self.device = model_runner.device
        self.max_bs = max_bs
        self.capture_bs = capture_bs or get_batch_sizes_to_capture(max_bs)
        dev = self.device
        self.fill = model_runner.attn_backend.get_cuda_graph_seq_len_fill_value()
        # static input buffers (decode_cuda_graph_runner.py input_buffers)
        self.input_ids = torch.zeros(max_bs, dtype=torch.int64, device=dev)
        self.req_pool_indices = torch.zeros(max_bs, dtype=torch.int64, device=dev)      # row 0 = padding row
        self.seq_lens = torch.full((max_bs,), self.fill, dtype=torch.int32, device=dev)
        self.out_cache_loc = torch.zeros(max_bs, dtype=torch.int64, device=dev)          # slot 0 = sink
        self.positions = torch.zeros(max_bs, dtype=torch.int64, device=dev)
        self.graphs: Dict[int, torch.cuda.CUDAGraph] = {}
        self.outputs: Dict[int, torch.Tensor] = {}
        self.pool = None
        model_runner.attn_backend.init_cuda_graph_state(max_bs, max_bs)
        self.capture()

    def _static_batch(self, bs: int) -> ForwardBatch:
        mr = self.mr
        return ForwardBatch(forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=self.input_ids[:bs],
                            req_pool_indices=self.req_pool_indices[:bs], seq_lens=self.seq_lens[:bs],
                            out_cache_loc=self.out_cache_loc[:bs], seq_lens_sum=bs * self.fill, seq_lens_cpu=None,
                            positions=self.positions[:bs], req_to_token_pool=mr.req_to_token_pool,
                            token_to_kv_pool=mr.token_to_kv_pool, attn_backend=mr.attn_backend)

    # The captured step ends with the GREEDY pick of every row written straight into the graph's own input-id buffer: a caller whose
    # batch is all-greedy (harness/engine.py) has nothing to launch between two replays but its slot bookkeeping -- the eager
    # arg-max behind a replay cost the launch plus ~9 us of graph-to-eager hand-over per step, and a copy of the ids into this
    # buffer.  A sampling batch ignores the pick (the caller writes its sampled ids over it); the logits are returned either way.
    greedy_ids_in_graph = True

    def _run(self, fb: ForwardBatch) -> torch.Tensor:
        kernels.clamp_position(fb.seq_lens, out=fb.positions)      # in-graph: positions = seq_lens - 1
        self.mr.attn_backend.init_forward_metadata_in_graph(fb)    # in-graph: shared-prefix plan of this step
        logits = self.mr.model.forward(fb.input_ids, fb.positions, fb).next_token_logits
        if self.greedy_ids_in_graph and logits.dtype in (torch.float32, torch.bfloat16) and logits.stride(-1) == 1:
            kernels.argmax(logits, out=fb.input_ids)               # sampler.py:133-141 on the step's own logits
        return logits

    def capture(self) -> None:
        # A Python GC pass that frees device tensors / graphs of an earlier runner while a capture
        # is open aborts the process (the reference freezes the GC for the same reason:
        # runner/decode_cuda_graph_runner.py freeze_gc).
        import gc

        gc.collect()
        torch.cuda.synchronize()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            self._capture_all()
        finally:
            if gc_was_enabled:
                gc.enable()

    def _capture_all(self) -> None:
        stream = torch.cuda.Stream(device=self.device)
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for bs in reversed(self.capture_bs):          # largest first so its pool is reused
                fb = self._static_batch(bs)
                self.mr.attn_backend.init_forward_metadata_out_graph(fb, in_capture=True)
                for _ in range(2):                        # warm-up runs (lazy hipBLASLt init etc.)
                    self._run(fb)
                torch.cuda.current_stream().synchronize()
                g = torch.cuda.CUDAGraph()
                # with collectives in the graph (TP > 1) RCCL's watchdog thread issues event queries while
                # the capture is open: only THIS thread's calls may invalidate it
                mode = "thread_local" if getattr(self.mr, "tp_size", 1) > 1 else "global"
                with torch.cuda.graph(g, pool=self.pool, stream=stream, capture_error_mode=mode):
                    out = self._run(fb)
                if self.pool is None:
                    self.pool = g.pool()
                self.graphs[bs] = g
                self.outputs[bs] = out
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()

    def can_run(self, bs: int) -> bool:
        return bs <= self.max_bs

    def replay(self, fb: ForwardBatch) -> LogitsProcessorOutput:
        raw = fb.batch_size
        bs = next(b for b in self.capture_bs if b >= raw)
        if bs != raw:                                     # pad: row 0 / slot 0 / fill length
            self.input_ids[raw:bs].zero_()
            self.req_pool_indices[raw:bs].zero_()
            self.seq_lens[raw:bs].fill_(self.fill)
            self.out_cache_loc[raw:bs].zero_()
        # a caller that keeps its batch state IN these buffers (harness/engine.py) has nothing to copy
        for buf, src in ((self.input_ids, fb.input_ids), (self.req_pool_indices, fb.req_pool_indices),
                         (self.seq_lens, fb.seq_lens), (self.out_cache_loc, fb.out_cache_loc)):
            if src.data_ptr() != buf.data_ptr():
                buf[:raw].copy_(src, non_blocking=True)
        self.graphs[bs].replay()
        return LogitsProcessorOutput(next_token_logits=self.outputs[bs][:raw])
