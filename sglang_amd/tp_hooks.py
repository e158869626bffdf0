"""Tensor parallelism through the drop-in surface: the reference's own TP groups carry the xGMI collectives.

Under an unchanged sglang TP > 1 launch every row-parallel projection ends in `GroupCoordinator.all_reduce`
(/root/reference/python/sglang/srt/distributed/parallel_state.py:648-758), the vocab-parallel logits in
`GroupCoordinator.all_gather` (:1273-1345), and the norm behind a projection may ask for the fused form through
`GroupCoordinator.fused_allreduce_rmsnorm` (:774-833, reached from layers/layernorm.py:191-245
`_forward_with_allreduce_fusion`).  `plugin.load()` registers AROUND hooks (srt/plugins/hook_registry.py:84,146) on those
four methods -- and on `GroupCoordinator.__init__`, where the communicator is built:

  __init__                 after the reference's own constructor: for a GPU group of 2 / 4 / 8 ranks created with
                           `use_custom_allreduce` (the reference's own hint: server_args --disable-custom-all-reduce turns
                           it off) an XgmiAllReduce is attached -- IPC handles exchanged over the group's OWN gloo group
                           (`cpu_group`), proved against the group's OWN RCCL group (`device_group`) by the start-up
                           self-test and dropped on EVERY rank together if any rank fails (parallel_state.start_xgmi);
                           every group gets its OWN communicator (workspace + flag counters), as the reference's `ca_comm`;
                           launches of one communicator from two streams are ordered by the communicator itself
                           (`XgmiAllReduce._ordered`, eagerly and inside captures)
  all_reduce               bf16 messages up to 64 MiB: one launch on the current stream (one-shot below the reference's
                           switch points, custom_all_reduce.py:260-307, two-stage above), graph-capturable; everything
                           else -- and every group without a communicator -- is the reference's method, untouched
  fused_allreduce_rmsnorm  (out, residual) = RMSNorm(all_reduce(x) + residual) in the all-reduce's epilogue, residual
                           updated in place as the reference's fused kernels do; None-returning fallbacks stay the
                           reference's
  all_gather               2-D bf16 shards along the last dimension (logits_processor.py:676): one launch

The fused decode layer (fused_decode.py) reads the communicator of the reference's TP group through `tp_communicator()`,
so a TP > 1 decode batch keeps its 9 launches per layer: the projection GEMM, then the all-reduce whose epilogue IS the
residual add + RMSNorm.
"""
from __future__ import annotations

import warnings
from typing import Optional

import torch

XGMI_ATTR = "_sgl_amd_xgmi"
# group names of srt/distributed/parallel_state.py:2451-2672 whose all-reduces sit on the decode path
GROUP_NAMES = ("tp", "attention_tp", "moe_tp", "moe_ep")
_COMMS = []               # every communicator built here (close_all)

_P = "sglang.srt.distributed.parallel_state.GroupCoordinator."
HOOK_TARGETS = (_P + "__init__", _P + "all_reduce", _P + "fused_allreduce_rmsnorm", _P + "all_gather")


def _group_base_name(group) -> str:
    name = str(getattr(group, "unique_name", ""))
    return name.rsplit(":", 1)[0]                  # parallel_state.py _get_unique_name: "<name>:<counter>"


def attach(group):
    """Build (or share) the xGMI communicator of a reference GroupCoordinator.  Collective over the group: every rank of
    it runs this from the same constructor call.  Returns the communicator or None; never raises."""
    setattr(group, XGMI_ATTR, None)
    try:
        world = int(group.world_size)
        dev = getattr(group, "device", None)
        if (world not in (2, 4, 8) or not getattr(group, "use_custom_allreduce", False) or dev is None
                or torch.device(dev).type != "cuda" or _group_base_name(group) not in GROUP_NAMES
                or getattr(group, "cpu_group", None) is None or getattr(group, "device_group", None) is None):
            return None
        # ONE communicator per GroupCoordinator -- its own workspace and flag counters, as the reference gives every group its own
        # `ca_comm` (parallel_state.py:405-470): groups over the same ranks (tp / attention_tp / moe_tp of a plain TP launch) may then
        # run on different streams, or in different branches of one capture, without sharing a byte.  (Rounds 4-5 shared one
        # communicator per rank set and serialised its launches; 128 MiB of workspace per group is the price of not doing that.)
        from .distributed.parallel_state import start_xgmi

        comm = start_xgmi(group.cpu_group, group.device_group, int(group.rank_in_group), world, torch.device(dev))
        if comm is not None:
            _COMMS.append(comm)
        setattr(group, XGMI_ATTR, comm)
        return comm
    except Exception as e:                         # noqa: BLE001 -- the reference's own collectives remain
        warnings.warn(f"sglang_amd: no xGMI communicator for group {getattr(group, 'unique_name', '?')} "
                      f"({type(e).__name__}: {e}); the reference's collectives are used")
        return None


def communicator_of(group):
    return getattr(group, XGMI_ATTR, None) if group is not None else None


def tp_communicator():
    """(tp_size, communicator) of the running job: the reference's TP group when running under it (its communicator is
    the one `attach` built), this package's own process group otherwise."""
    try:
        from sglang.srt.distributed.parallel_state import get_tp_group

        g = get_tp_group()
        return int(g.world_size), communicator_of(g)
    except Exception:
        from .distributed import parallel_state as ps

        return ps.get_tensor_model_parallel_world_size(), ps.get_xgmi_all_reduce()


def _takes(comm, x: torch.Tensor) -> bool:
    return comm is not None and isinstance(x, torch.Tensor) and (comm.should_use(x) or comm.should_use_two_stage(x))


# ---- the hooks: HookType.AROUND = hook(original_fn, *args, **kwargs) ------------------------------------------------
def group_init_hook(original, self, *args, **kwargs):
    original(self, *args, **kwargs)
    attach(self)


def group_all_reduce_hook(original, self, input_):
    comm = communicator_of(self)
    if comm is not None and self.world_size > 1 and not torch.compiler.is_compiling() and _takes(comm, input_):
        return comm.all_reduce_any(input_)
    return original(self, input_)


def group_fused_allreduce_rmsnorm_hook(original, self, input_, residual_inp_, weight_, eps):
    comm = communicator_of(self)
    if (comm is not None and self.world_size > 1 and not torch.compiler.is_compiling() and _takes(comm, input_)
            and input_.dim() == 2 and input_.shape[-1] <= 16384 and isinstance(residual_inp_, torch.Tensor)
            and residual_inp_.shape == input_.shape and residual_inp_.is_contiguous() and residual_inp_.dtype == input_.dtype
            and weight_.dtype == input_.dtype):
        # `residual_inp_` is updated IN PLACE and handed back as the residual output: the one caller
        # (layernorm.py:198-245 `_forward_with_allreduce_fusion`) returns the pair as its (hidden, residual) and never reads
        # its own `residual` again -- a caller that kept using the tensor it passed in would see it changed
        out = comm.all_reduce_add_rmsnorm(input_, residual_inp_, weight_, float(eps))
        return out, residual_inp_
    return original(self, input_, residual_inp_, weight_, eps)


def group_all_gather_hook(original, self, input_, dim=-1, output_tensor_list=None):
    comm = communicator_of(self)
    if (comm is not None and self.world_size > 1 and output_tensor_list is None and not torch.compiler.is_compiling()
            and isinstance(input_, torch.Tensor) and input_.is_cuda and input_.dim() == 2 and dim in (-1, 1)
            and input_.dtype == torch.bfloat16 and input_.shape[1] % 8 == 0 and not comm.disabled
            and comm.fits_all_gather(input_)):
        return comm.all_gather(input_.contiguous())
    return original(self, input_, dim, output_tensor_list)


_HOOKS = (group_init_hook, group_all_reduce_hook, group_fused_allreduce_rmsnorm_hook, group_all_gather_hook)


def install(registry, hook_type_around) -> None:
    """plugin.load(): HookRegistry.register(target, hook, HookType.AROUND) for the four GroupCoordinator methods."""
    for target, hook in zip(HOOK_TARGETS, _HOOKS):
        if not any(h is hook for _, h, _ in registry._hooks.get(target, [])):
            registry.register(target, hook, hook_type_around)


def close_all() -> None:
    """Process shutdown / tests: unmap and free every communicator built here."""
    for comm in list(_COMMS):
        try:
            comm.close()
        except Exception:
            pass
    _COMMS.clear()
