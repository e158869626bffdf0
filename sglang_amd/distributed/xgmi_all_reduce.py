"""One-shot all-reduce communicator over xGMI peer mappings (host side of csrc/all_reduce.hip).

Takes the place of `CustomAllreduce` (/root/reference/python/sglang/srt/distributed/device_communicators/
custom_all_reduce.py:40-340: buffer creation + IPC handle exchange :182-258, `should_custom_ar` :260-290, the
dispatch :292-340) under `GroupCoordinator.all_reduce` (srt/distributed/parallel_state.py:648-758).

One workspace per communicator and rank (32 KiB of flags + a data area for the largest message), allocated by the library as its own
uncached hipMalloc so that it can be exported; the 64-byte hipIpcMemHandles travel through the (CPU / gloo or RCCL)
process group once, every rank maps every peer's workspace, and from then on a call is ONE kernel launch on the
current stream with no host state -- it records into the decode hipGraph like any other kernel, no
"graph buffer registration" pass (custom_all_reduce.py:182-258) is needed because the input is copied into the
registered buffer by the kernel itself (<= 2 MiB, on-chip speed).

Messages above `max_bytes` (prefill activations) are left to RCCL.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import native

DEFAULT_MAX_BYTES = 2 * 1024 * 1024       # the one-shot regime (custom_all_reduce.py: max_size for the 1-stage kernel)
DEFAULT_TWO_STAGE_BYTES = 64 * 1024 * 1024  # largest message of the two-stage kernel (a prefill chunk's [T, hidden] activations)


def one_shot_limit(world: int, max_bytes: int) -> int:
    """custom_all_reduce.py:260-307: two ranks always pull the whole message; with more ranks the one-shot kernel moves
    (world - 1) copies per rank where the two-stage one moves 2 (world - 1) / world, at the price of a third flag
    barrier (~2 us): 512 KiB at world 4, 256 KiB at world 8."""
    return max_bytes if world == 2 else min(max_bytes, 512 * 1024 if world == 4 else 256 * 1024)


class _LaunchOrder:
    def __init__(self, comm):
        self.comm = comm

    def __enter__(self):
        c = self.comm
        self.cur = torch.cuda.current_stream()
        self.capturing = torch.cuda.is_current_stream_capturing()
        last = c._last_stream
        if last is not None and last != self.cur and c._last_captured == self.capturing:
            if self.capturing:
                # inside ONE capture the previous launch sits on another capturing stream: a fresh event recorded there is the edge
                ev = torch.cuda.Event()
                ev.record(last)
                self.cur.wait_event(ev)
            elif c._last_event is not None:
                self.cur.wait_event(c._last_event)
        return self

    def __exit__(self, *exc):
        c = self.comm
        if not self.capturing:
            if c._last_event is None:
                c._last_event = torch.cuda.Event()
            c._last_event.record(self.cur)
        c._last_stream, c._last_captured = self.cur, self.capturing
        return False


class XgmiAllReduce:
    def __init__(self, group, rank: int, world: int, device: torch.device, max_bytes: int = DEFAULT_MAX_BYTES,
                 handle_exchange=None, two_stage_bytes: int = DEFAULT_TWO_STAGE_BYTES, policy_world: Optional[int] = None):
        """`group`: a process group every rank of the TP group is in (used once, for the handle exchange; may be a
        gloo group).  `handle_exchange(bytes) -> List[bytes]` overrides the collective (tests).  `policy_world`: the
        TP degree whose one-shot / two-stage switch point applies (a world-of-1 loopback communicator standing in for one
        rank of a TP job launches the kernels that job would)."""
        lib = native.lib()
        if world not in (1, 2, 4, 8) or world > lib.sgl_amd_xgmi_max_world():
            raise ValueError(f"XgmiAllReduce: world size {world} (supported: 2, 4, 8; 1 = loopback for rank-shape runs)")
        self.rank, self.world, self.device = rank, world, device
        self.policy_world = int(policy_world or world)
        self.max_bytes = int(max_bytes)
        self.two_stage_bytes = int(two_stage_bytes)
        self.data_offset = int(lib.sgl_amd_xgmi_data_offset())
        # data area: the one-shot message, or the two halves (copies + published sums) of a two-stage message
        self.ws_bytes = int(lib.sgl_amd_xgmi_workspace_bytes(max(self.max_bytes, 2 * self.two_stage_bytes + 512)))
        torch.cuda.set_device(device)
        self._own, self._opened = None, []
        hbytes = lib.sgl_amd_xgmi_ipc_handle_bytes()
        # Local steps first, under a guard: a rank whose allocation / export fails must STILL take part in the handle
        # exchange (with a None marker), or its peers would sit in the collective while it moves on to the next one.
        mine, local_error = None, None
        try:
            ptr = ctypes.c_void_p()
            native.call("sgl_amd_xgmi_alloc", self.ws_bytes, ctypes.byref(ptr))
            self._own = ptr.value
            # SGLANG_AMD_XGMI_RELEASE_FENCE=1: the fallback protocol (a system-scope release fence ahead of every flag) for every
            # communicator this process builds; the setting itself lives in the communicator's own signal block
            if os.environ.get("SGLANG_AMD_XGMI_RELEASE_FENCE", "") not in ("", "0"):
                native.call("sgl_amd_xgmi_set_release_fence", self._own, 1)
            buf = ctypes.create_string_buffer(hbytes)
            native.call("sgl_amd_xgmi_ipc_get_handle", self._own, buf)
            mine = bytes(buf.raw)
        except Exception as e:                    # noqa: BLE001 -- reported after the exchange
            local_error = e
        if handle_exchange is not None:
            handles = handle_exchange(mine)
        else:
            handles: List[Optional[bytes]] = [None] * world
            dist.all_gather_object(handles, mine, group=group)
        try:
            if local_error is not None:
                raise local_error
            if any(h is None for h in handles):
                raise RuntimeError(f"XgmiAllReduce: rank(s) {[r for r, h in enumerate(handles) if h is None]} could not export a workspace")
            peers = (ctypes.c_void_p * world)()
            for r in range(world):
                if r == rank:
                    peers[r] = self._own
                else:
                    p = ctypes.c_void_p()
                    native.call("sgl_amd_xgmi_ipc_open_handle", ctypes.create_string_buffer(handles[r], hbytes), ctypes.byref(p))
                    peers[r] = p.value
                    self._opened.append(p.value)
        except Exception:
            self.close()                          # unmap what was opened, free the workspace: nothing leaks on the error path
            raise
        self._peers = peers                      # host array of device pointers (kept alive with the object)
        self.disabled = False
        self._last_stream, self._last_event, self._last_captured = None, None, False

    def _ordered(self):
        """Stream ordering of THIS communicator's launches (every launch method runs under it, whoever the caller is: the
        GroupCoordinator hooks, the fused decode layer, tests).  One data area and one set of monotonically increasing flag
        counters: its launches are only correct one after the other.  On one stream that is program order; a launch arriving on
        ANOTHER stream than the previous one first waits for an event recorded behind the previous launch -- eagerly, and inside a
        stream capture too (alt-stream branches forked inside a captured forward: the event is recorded on the previous capturing
        stream and becomes a graph edge).  An eager launch is never made to wait on a launch that was only captured."""
        return _LaunchOrder(self)

    # custom_all_reduce.py:260-290 should_custom_ar
    def should_use(self, x: torch.Tensor) -> bool:
        return (not self.disabled and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() >= 1
                and x.shape[-1] % 8 == 0 and 0 < x.numel() * 2 <= self.max_bytes)

    def all_reduce(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, *, residual: Optional[torch.Tensor] = None,
                   norm_weight: Optional[torch.Tensor] = None, eps: float = 0.0, num_blocks: int = 0) -> torch.Tensor:
        """out = sum over ranks of x (fp32 accumulation in rank order, one rounding: identical bits on every rank).
        With residual / norm_weight: residual <- bf16(out + residual) in place, returns RMSNorm(residual)."""
        if not self.should_use(x):
            raise ValueError("XgmiAllReduce.all_reduce: tensor outside the one-shot regime (bf16, contiguous, "
                             f"last dim % 8 == 0, <= {self.max_bytes} bytes)")
        hidden = x.shape[-1]
        rows = x.numel() // hidden
        if out is None:
            out = torch.empty_like(x)
        epilogue = 0
        if residual is not None:
            if norm_weight is None or residual.shape != x.shape or not residual.is_contiguous():
                raise ValueError("XgmiAllReduce.all_reduce: add_rmsnorm needs a contiguous residual of x's shape and norm_weight")
            epilogue = 1
        with self._ordered():
            native.call("sgl_amd_xgmi_one_shot_all_reduce", x.data_ptr(), out.data_ptr(), rows, hidden, self.rank, self.world,
                        self._peers, self.ws_bytes, epilogue, residual.data_ptr() if residual is not None else None,
                        norm_weight.data_ptr() if norm_weight is not None else None, float(eps), int(num_blocks),
                        torch.cuda.current_stream().cuda_stream)
        return out

    def should_use_two_stage(self, x: torch.Tensor) -> bool:
        return (not self.disabled and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() >= 1
                and x.shape[-1] % 8 == 0 and 0 < x.numel() * 2 <= self.two_stage_bytes)

    def all_reduce_any(self, x: torch.Tensor) -> torch.Tensor:
        """The reference's dispatch (custom_all_reduce.py:292-340): one-shot below the size where link traffic starts to
        dominate, two-stage above."""
        if self.should_use(x) and x.numel() * 2 <= one_shot_limit(self.policy_world, self.max_bytes):
            return self.all_reduce(x)
        return self.two_stage_all_reduce(x)

    def two_stage_all_reduce(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, num_blocks: int = 0, *,
                             residual: Optional[torch.Tensor] = None, norm_weight: Optional[torch.Tensor] = None,
                             eps: float = 0.0) -> torch.Tensor:
        if not self.should_use_two_stage(x) or x.shape[-1] % 8 != 0:
            raise ValueError(f"XgmiAllReduce.two_stage_all_reduce: bf16, contiguous, last dim % 8 == 0, <= {self.two_stage_bytes} bytes")
        if out is None:
            out = torch.empty_like(x)
        hidden = x.shape[-1]
        epilogue = 0
        if residual is not None:
            if norm_weight is None or residual.shape != x.shape or not residual.is_contiguous() or hidden > 16384:
                raise ValueError("XgmiAllReduce.two_stage_all_reduce: add_rmsnorm needs a contiguous residual of x's shape, norm_weight, hidden <= 16384")
            epilogue = 1
        with self._ordered():
            native.call("sgl_amd_xgmi_two_stage_all_reduce", x.data_ptr(), out.data_ptr(), x.numel() // hidden, hidden, self.rank, self.world,
                        self._peers, self.ws_bytes, epilogue, residual.data_ptr() if residual is not None else None,
                        norm_weight.data_ptr() if norm_weight is not None else None, float(eps), int(num_blocks),
                        torch.cuda.current_stream().cuda_stream)
        return out

    def all_reduce_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, norm_weight: torch.Tensor, eps: float) -> torch.Tensor:
        """RMSNorm(all_reduce(x) + residual) in ONE launch whatever the size class (residual updated in place)."""
        if self.should_use(x) and x.numel() * 2 <= one_shot_limit(self.policy_world, self.max_bytes):
            return self.all_reduce(x, residual=residual, norm_weight=norm_weight, eps=eps)
        return self.two_stage_all_reduce(x, residual=residual, norm_weight=norm_weight, eps=eps)

    def fits_all_gather(self, x: torch.Tensor) -> bool:
        return self.data_offset + x.numel() * 2 <= self.ws_bytes

    def set_release_fence(self, on: bool) -> None:
        """Fallback switch of THIS communicator's flag protocol (include/sglang_amd.h: sgl_amd_xgmi_set_release_fence): a word of
        its own signal block, read by the kernels when a launch runs (captured graphs follow it)."""
        torch.cuda.synchronize(self.device)
        native.call("sgl_amd_xgmi_set_release_fence", self._own, 1 if on else 0)

    def all_gather(self, x: torch.Tensor, num_blocks: int = 0) -> torch.Tensor:
        """[rows, cols] per rank -> [rows, world * cols] (tensor_model_parallel_all_gather(dim=-1) of a 2-D tensor)."""
        if not (x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 2 and x.shape[1] % 8 == 0
                and self.fits_all_gather(x)):
            raise ValueError("XgmiAllReduce.all_gather: 2-D contiguous bf16 shard with cols % 8 == 0 that fits the workspace")
        out = torch.empty((x.shape[0], x.shape[1] * self.world), dtype=x.dtype, device=x.device)
        with self._ordered():
            native.call("sgl_amd_xgmi_all_gather", x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], self.rank, self.world, self._peers,
                        self.ws_bytes, int(num_blocks), torch.cuda.current_stream().cuda_stream)
        return out

    def arm(self, trap_on_timeout: bool = True) -> None:
        """After the start-up self-test: a flag wait that gives up traps the kernel instead of returning garbage."""
        native.call("sgl_amd_xgmi_arm", self._own, 1 if trap_on_timeout else 0)

    def timed_out(self) -> bool:
        """True when a flag wait gave up because a peer never arrived (synchronises the device)."""
        return bool(native.lib().sgl_amd_xgmi_timed_out(self._own))

    def close(self) -> None:
        for p in self._opened:
            native.call("sgl_amd_xgmi_ipc_close_handle", p)
        self._opened = []
        if self._own:
            native.call("sgl_amd_xgmi_free", self._own)
            self._own = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
