"""Tensor-parallel process group: one process per GPU, RCCL over xGMI
(`torch.distributed` backend "nccl" IS RCCL on ROCm); gloo on CPU for tests.

Mirrors the small part of /root/reference/python/sglang/srt/distributed/
parallel_state.py (:237 GroupCoordinator, :648-758 all_reduce) and
communication_op.py:18-107 that the hot path calls.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

_TP_GROUP = None
_TP_CPU_GROUP = None      # gloo group of the same ranks: control-plane exchanges (IPC handles), like the reference's cpu_group
_TP_SIZE = 1
_TP_RANK = 0
_XGMI = None              # one-shot all-reduce communicator (decode-sized messages), None = RCCL only
_EMULATED = None          # (rank, size): ONE rank of a TP job on one GPU, collectives = world-of-1 loopback launches


def emulate_tensor_parallel_rank(rank: int, size: int, device) -> None:
    """Rank-shape runs (bench.py --rank-of, tests): this process plays rank `rank` of a TP=`size` job on ONE GPU.
    Models build that rank's weight shards and take the TP>1 code paths; every collective is the xGMI kernel of the
    real job launched over a world of one (copy into the workspace, flag barriers, the sum of one copy, the fused
    epilogue -- everything but the wire), so the launches sit in the decode graph where the real ones would.
    Messages outside the kernels' sizes pass through unchanged.  NOT a multi-GPU measurement."""
    global _TP_SIZE, _TP_RANK, _XGMI, _EMULATED
    from .xgmi_all_reduce import XgmiAllReduce

    _TP_SIZE, _TP_RANK, _EMULATED = size, rank, (rank, size)
    _XGMI = XgmiAllReduce(None, 0, 1, device, handle_exchange=lambda h: [h], policy_world=size)


def init_distributed_environment(backend: Optional[str] = None, tp_size: Optional[int] = None,
                                 device_index: Optional[int] = None, xgmi_all_reduce: Optional[bool] = None) -> None:
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).  `device_index` overrides LOCAL_RANK
    (tests put two ranks on one GPU); `xgmi_all_reduce` forces the one-shot communicator on / off (default: on for
    GPU groups of 2 / 4 / 8 ranks unless SGLANG_AMD_XGMI_AR=0)."""
    global _TP_GROUP, _TP_SIZE, _TP_RANK, _TP_CPU_GROUP, _XGMI
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        _TP_GROUP, _TP_SIZE, _TP_RANK = None, 1, 0
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(device_index if device_index is not None else int(os.environ.get("LOCAL_RANK", rank)))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    _TP_SIZE = tp_size or world
    assert world % _TP_SIZE == 0
    # consecutive ranks form a TP group (one node: all of them)
    for g0 in range(0, world, _TP_SIZE):
        ranks = list(range(g0, g0 + _TP_SIZE))
        grp = dist.new_group(ranks) if _TP_SIZE != world else dist.group.WORLD
        cpu_grp = dist.new_group(ranks, backend="gloo") if backend != "gloo" else grp
        if rank in ranks:
            _TP_GROUP = grp
            _TP_CPU_GROUP = cpu_grp
            _TP_RANK = rank - g0
    if xgmi_all_reduce is None:
        xgmi_all_reduce = (torch.cuda.is_available() and _TP_SIZE in (2, 4, 8)
                           and os.environ.get("SGLANG_AMD_XGMI_AR", "1") != "0")
    if xgmi_all_reduce:
        _XGMI = start_xgmi(_TP_CPU_GROUP, _TP_GROUP, _TP_RANK, _TP_SIZE, torch.device("cuda", torch.cuda.current_device()))


def start_xgmi(cpu_group, device_group, rank: int, size: int, dev):
    """`cpu_group` / `device_group`: the gloo and the RCCL process group of the SAME `size` ranks (`rank` = this process's
    index in them) -- this package's own TP group, or the reference's GroupCoordinator.cpu_group / .device_group
    (tp_hooks.attach).  Create the xGMI communicator and PROVE it on this node before anything depends on it: every rank runs a known
    tensor through each of its kernels (one-shot, two-stage, all-gather) and through the group's own collectives; unless all ranks see identical,
    correct results (and no flag wait gave up) the communicator is dropped on EVERY rank and RCCL carries all
    all-reduces -- the reference degrades the same way when its custom all-reduce cannot be set up
    (custom_all_reduce.py:100-180).  A rank-local failure must not leave the ranks with different algorithms."""
    import warnings

    from .xgmi_all_reduce import XgmiAllReduce

    xg, ok = None, 1
    try:
        xg = XgmiAllReduce(cpu_group, rank, size, dev)
    except Exception as e:                      # e.g. IPC not permitted between these devices
        warnings.warn(f"one-shot xGMI all-reduce unavailable ({type(e).__name__}: {e}); using RCCL")
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=cpu_group)
    if int(flag) == 1:
        try:
            g = torch.Generator(device="cpu").manual_seed(1234 + rank)
            good = True
            # three rounds on fresh data through the SAME workspaces: a stale line of a peer's workspace (the one failure
            # the uncached mappings and the acquire fence must exclude) would show from the second round on
            for _ in range(3):
                x = (torch.randn((64, 4096), generator=g) * 0.5).to(torch.bfloat16).to(dev)
                mine = xg.all_reduce(x.clone())
                ref = x.float().clone()
                dist.all_reduce(ref, group=device_group)                   # fp32 sum through the group's collective
                # ... and the other two kernels the decode graph may hold (weak-scaling batches put 256 / 512 rows on a
                # rank: two-stage all-reduce; vocab-parallel logits: all-gather), on shapes with ragged row chunks
                y = (torch.randn((301, 1024), generator=g) * 0.5).to(torch.bfloat16).to(dev)
                mine2 = xg.two_stage_all_reduce(y.clone())
                ref2 = y.float().clone()
                dist.all_reduce(ref2, group=device_group)
                z = (torch.randn((17, 256), generator=g) * 0.5).to(torch.bfloat16).to(dev)
                mine3 = xg.all_gather(z)
                parts = [torch.empty_like(z) for _ in range(size)]
                dist.all_gather(parts, z, group=device_group)
                torch.cuda.synchronize()
                good = good and (not xg.timed_out()) and bool(((mine.float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 1e-2).all()) \
                    and bool(((mine2.float() - ref2).abs() <= 2.0 ** -7 * ref2.abs() + 1e-2).all()) \
                    and bool(torch.equal(mine3, torch.cat(parts, dim=1)))
        except Exception as e:
            warnings.warn(f"one-shot xGMI all-reduce self-test raised {type(e).__name__}: {e}")
            good = False
        flag = torch.tensor([1 if good else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=cpu_group)
    if int(flag) == 1:
        # from here on an all-reduce that cannot complete must not hand back an unreduced sum: a flag wait that gives up
        # traps the kernel (the stream fails, the next sync raises on this rank; its peers time out the same way)
        xg.arm(True)
        return xg
    if xg is not None:
        warnings.warn("one-shot xGMI all-reduce failed its start-up self-test on some rank; using RCCL for every all-reduce")
        try:
            xg.close()
        except Exception:
            pass
    return None


def destroy() -> None:
    global _TP_GROUP, _TP_SIZE, _TP_RANK, _TP_CPU_GROUP, _XGMI, _EMULATED
    _EMULATED = None
    if _XGMI is not None:
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier(group=_TP_CPU_GROUP)        # nobody unmaps a workspace a peer's kernel may still read
        _XGMI.close()
        _XGMI = None
    if dist.is_initialized():
        dist.destroy_process_group()
    _TP_GROUP, _TP_CPU_GROUP, _TP_SIZE, _TP_RANK = None, None, 1, 0


def get_xgmi_all_reduce():
    return _XGMI


def get_tensor_model_parallel_world_size() -> int:
    return _TP_SIZE


def get_tensor_model_parallel_rank() -> int:
    return _TP_RANK


def get_tp_group():
    return _TP_GROUP


def tensor_model_parallel_all_reduce(x: torch.Tensor) -> torch.Tensor:
    """SUM over the TP ranks (communication_op.py:18, parallel_state.py:648-758 dispatch): decode-sized bf16
    messages take the one-shot xGMI kernel (one launch on the current stream, graph-capturable), everything else
    RCCL (which enqueues on its own stream behind an event on the current one)."""
    if _TP_SIZE == 1:
        return x
    if _XGMI is not None and (_XGMI.should_use(x) or _XGMI.should_use_two_stage(x)):
        return _XGMI.all_reduce_any(x)
    if _EMULATED is not None:
        return x
    dist.all_reduce(x, group=_TP_GROUP)
    return x


def tensor_model_parallel_all_reduce_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, norm_weight: torch.Tensor,
                                                 eps: float) -> torch.Tensor:
    """RMSNorm(all_reduce(x), residual) -- the pair that follows every row-parallel projection of a decoder layer
    (llama.py:341-370 through LayerCommunicator).  residual is updated in place, the normed activations returned.
    With the one-shot communicator this is ONE kernel (sum + residual add + norm in the all-reduce's epilogue)."""
    from .. import kernels

    if (_TP_SIZE > 1 and _XGMI is not None and residual.is_contiguous() and x.shape[-1] <= 16384
            and (_XGMI.should_use(x) or _XGMI.should_use_two_stage(x))):
        return _XGMI.all_reduce_add_rmsnorm(x, residual, norm_weight, eps)
    x = tensor_model_parallel_all_reduce(x)
    kernels.fused_add_rmsnorm(x, residual, norm_weight, eps)
    return x


# rows from which a row-parallel projection takes the piecewise form below (SGLANG_AMD_PIECEWISE_MIN_ROWS overrides: a dry run whose
# ranks share one GPU over gloo groups keeps everything on the xGMI kernels -- gloo's staged device all-reduce and a spinning
# xGMI launch of the other rank time-slicing the same GPU wait for each other)
PIECEWISE_MIN_ROWS = int(os.environ.get("SGLANG_AMD_PIECEWISE_MIN_ROWS", "2048"))


def row_parallel_linear(x: torch.Tensor, weight: torch.Tensor, min_rows_per_chunk: int = 1024, max_chunks: int = 4
                        ) -> torch.Tensor:
    """all_reduce(x @ weight^T) for a prefill-sized row-parallel projection with the collective OVERLAPPED with the
    matmul: the rows are cut into up to `max_chunks` pieces, piece i's RCCL all-reduce runs on RCCL's own HIP stream
    (async_op) while piece i+1's GEMM runs on the compute stream; the compute stream waits for the collectives only
    at the end.  xGMI is point-to-point (7 links per GPU), so a 100 MB all-reduce is link-bound for about as long as
    the GEMM that produced it takes -- hiding it is worth one extra GEMM launch per piece."""
    rows = x.shape[0]
    n = min(max_chunks, rows // max(1, min_rows_per_chunk))
    if _TP_SIZE == 1 or n <= 1 or _EMULATED is not None:
        return tensor_model_parallel_all_reduce(torch.nn.functional.linear(x, weight))
    out = torch.empty((rows, weight.shape[0]), dtype=x.dtype, device=x.device)
    step = (rows + n - 1) // n
    works = []
    for a in range(0, rows, step):
        b = min(rows, a + step)
        torch.matmul(x[a:b], weight.t(), out=out[a:b])
        works.append(dist.all_reduce(out[a:b], group=_TP_GROUP, async_op=True))
    for w in works:
        w.wait()
    return out


def tensor_model_parallel_all_gather(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    if _TP_SIZE == 1:
        return x
    if dim < 0:
        dim += x.dim()
    if (_XGMI is not None and not _XGMI.disabled and x.is_cuda and x.dim() == 2 and dim == 1 and x.dtype == torch.bfloat16
            and x.shape[1] % 8 == 0 and _XGMI.fits_all_gather(x)):
        return _XGMI.all_gather(x.contiguous())          # one launch on the current stream: the decode graph holds no RCCL node
    if _EMULATED is not None:
        return x                                        # (loopback: the rank's own shard)
    if x.is_cuda and dist.get_backend(_TP_GROUP) == "gloo":
        # gloo moves device tensors only for broadcast / all_reduce: stage through the host (two ranks on one GPU in
        # the single-device tests; a real node runs RCCL)
        xc = x.cpu()
        parts = [torch.empty_like(xc) for _ in range(_TP_SIZE)]
        dist.all_gather(parts, xc.contiguous(), group=_TP_GROUP)
        return torch.cat(parts, dim=dim).to(x.device)
    parts = [torch.empty_like(x) for _ in range(_TP_SIZE)]
    dist.all_gather(parts, x.contiguous(), group=_TP_GROUP)
    return torch.cat(parts, dim=dim)


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()
