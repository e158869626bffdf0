"""One-shot all-reduce communicator over xGMI peer mappings (host side of csrc/all_reduce.hip).

Takes the place of `CustomAllreduce` (/root/reference/python/sglang/srt/distributed/device_communicators/
custom_all_reduce.py:40-340: buffer creation + IPC handle exchange :182-258, `should_custom_ar` :260-290, the
dispatch :292-340) under `GroupCoordinator.all_reduce` (srt/distributed/parallel_state.py:648-758).

One workspace per rank (8 KiB of flags + a data area for the largest message), allocated by the library as its own
uncached hipMalloc so that it can be exported; the 64-byte hipIpcMemHandles travel through the (CPU / gloo or RCCL)
process group once, every rank maps every peer's workspace, and from then on a call is ONE kernel launch on the
current stream with no host state -- it records into the decode hipGraph like any other kernel, no
"graph buffer registration" pass (custom_all_reduce.py:182-258) is needed because the input is copied into the
registered buffer by the kernel itself (<= 2 MiB, on-chip speed).

Messages above `max_bytes` (prefill activations) are left to RCCL.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import native

DEFAULT_MAX_BYTES = 2 * 1024 * 1024       # the one-shot regime (custom_all_reduce.py: max_size for the 1-stage kernel)


class XgmiAllReduce:
    def __init__(self, group, rank: int, world: int, device: torch.device, max_bytes: int = DEFAULT_MAX_BYTES,
                 handle_exchange=None):
        """`group`: a process group every rank of the TP group is in (used once, for the handle exchange; may be a
        gloo group).  `handle_exchange(bytes) -> List[bytes]` overrides the collective (tests)."""
        lib = native.lib()
        if world not in (2, 4, 8) or world > lib.sgl_amd_xgmi_max_world():
            raise ValueError(f"XgmiAllReduce: world size {world} (supported: 2, 4, 8)")
        self.rank, self.world, self.device = rank, world, device
        self.max_bytes = int(max_bytes)
        self.ws_bytes = int(lib.sgl_amd_xgmi_workspace_bytes(self.max_bytes))
        torch.cuda.set_device(device)
        ptr = ctypes.c_void_p()
        native.call("sgl_amd_xgmi_alloc", self.ws_bytes, ctypes.byref(ptr))
        self._own = ptr.value
        hbytes = lib.sgl_amd_xgmi_ipc_handle_bytes()
        buf = ctypes.create_string_buffer(hbytes)
        native.call("sgl_amd_xgmi_ipc_get_handle", self._own, buf)
        mine = bytes(buf.raw)
        if handle_exchange is not None:
            handles = handle_exchange(mine)
        else:
            handles: List[Optional[bytes]] = [None] * world
            dist.all_gather_object(handles, mine, group=group)
        self._opened: List[int] = []
        peers = (ctypes.c_void_p * world)()
        for r in range(world):
            if r == rank:
                peers[r] = self._own
            else:
                p = ctypes.c_void_p()
                native.call("sgl_amd_xgmi_ipc_open_handle", ctypes.create_string_buffer(handles[r], hbytes), ctypes.byref(p))
                peers[r] = p.value
                self._opened.append(p.value)
        self._peers = peers                      # host array of device pointers (kept alive with the object)
        self.disabled = False

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
        native.call("sgl_amd_xgmi_one_shot_all_reduce", x.data_ptr(), out.data_ptr(), rows, hidden, self.rank, self.world,
                    self._peers, self.ws_bytes, epilogue, residual.data_ptr() if residual is not None else None,
                    norm_weight.data_ptr() if norm_weight is not None else None, float(eps), int(num_blocks),
                    torch.cuda.current_stream().cuda_stream)
        return out

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
