"""The REAL reference classes the drop-in boundary is handed under sglang -- `ReqToTokenPool`, `MHATokenToKVPool`
(srt/mem_cache/memory_pool.py), `RadixAttention` (srt/layers/radix_attention.py), `ForwardBatch` / `ForwardMode`
(srt/model_executor/forward_batch_info.py), `RadixCache` (srt/mem_cache/radix_cache.py), `RMSNorm`, `RotaryEmbedding`
-- importable WITHOUT an sglang install, through gen_golden.py's import hook (absent third-party packages stubbed,
modules that cannot import degrade to permissive stubs, the modules listed in REAL must import for real).

Test infrastructure only.  Two steps, because /root/reference does not exist on the GPU box:

    python tests/golden/ref_objects.py --stage     # build container: imports the classes from /root/reference, then copies
                                                   # exactly the reference files that were executed (a few hundred .py files)
                                                   # to oracle/_ref/sglang_objects/ -- git-ignored, never committed, travels
                                                   # with the gpurun snapshot like the staged Triton kernels
    pytest tests/test_reference_objects_gpu.py     # GPU box: imports them from the staged copy (skips when it is absent)

Every line of reference code that runs is the reference's own; nothing here restates it.
"""
from __future__ import annotations

import argparse
import importlib
import shutil
import sys
from contextlib import contextmanager
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
STAGE = REPO / "oracle" / "_ref" / "sglang_objects"
CONTAINER_REF = Path("/root/reference/python")
MODULES = ["sglang.srt.utils.common", "sglang.srt.mem_cache.memory_pool", "sglang.srt.layers.radix_attention",
           "sglang.srt.model_executor.forward_batch_info", "sglang.srt.mem_cache.radix_cache", "sglang.srt.runtime_context",
           "sglang.kernels.fused_op", "sglang.srt.layers.layernorm", "sglang.srt.layers.rotary_embedding.base",
           "sglang.srt.model_executor.forward_context",
           "sglang.srt.model_executor.runner_backend_utils.tc_piecewise_cuda_graph.context_manager"]
_installed = {}


def ref_root():
    """Where the reference sources can be imported from here: the container's checkout, else the staged copy, else None."""
    import os

    forced = os.environ.get("REF_OBJECTS_ROOT")          # tests of the staged copy inside the build container
    if forced:
        return Path(forced) if (Path(forced) / "sglang").exists() else None
    if (CONTAINER_REF / "sglang").exists():
        return CONTAINER_REF
    if (STAGE / "sglang").exists():
        return STAGE
    return None


def install(root: Path):
    """Install the import hook on `root` (once per process) and import the classes.  Returns a namespace of modules."""
    if _installed:
        return _installed["ns"]
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import gen_golden as G

    G.REF = Path(root)
    # `class X(msgspec.Struct, frozen=True)` with msgspec stubbed: the stub base must accept class keywords, or
    # sglang/kernels/fused_op.py (BaseFusedOp, the base of RMSNorm / RotaryEmbedding) cannot import for real
    G._Any.__init_subclass__ = classmethod(lambda cls, **kw: None)
    G.MISSING |= {"torchvision", "decord", "triton_kernels"}
    # (the hook skips package __init__ files; these two packages re-export the context accessors RadixAttention.forward calls)
    G.REAL |= set(MODULES) | {"sglang.srt.utils.custom_op", "sglang.srt.model_executor.runner_backend_utils.tc_piecewise_cuda_graph",
                              "sglang.srt.model_executor.runner_backend_utils.breakable_cuda_graph"}
    G.install_hook()
    ns = {m.rsplit(".", 1)[1]: importlib.import_module(m) for m in MODULES}
    ns["rotary_base"] = ns.pop("base")
    import types

    _installed["ns"] = types.SimpleNamespace(**ns, hook=G)
    return _installed["ns"]


@contextmanager
def single_rank(ns):
    """The reference's parallel context forced to one rank everywhere (runtime_context.py:155 override): no process group
    exists in a unit test, and ForwardBatch.init_new / the backends read the sizes through get_parallel()."""
    rt = ns.runtime_context
    ov = {f: (1 if f.endswith("size") else 0) for f in rt._PARALLEL_FIELDS if f.endswith(("size", "rank"))}
    with rt.get_parallel().override(**ov):
        yield


def stage() -> None:
    if not (CONTAINER_REF / "sglang").exists():
        raise SystemExit("/root/reference not present: staging only works in the build container")
    ns = install(CONTAINER_REF)
    failed = {name for name, _ in ns.hook.FAILED}
    files = []
    for name, mod in list(sys.modules.items()):
        f = getattr(mod, "__file__", None)
        if not name.startswith("sglang") or not f or name in failed or isinstance(mod, ns.hook._Stub):
            continue
        f = Path(f)
        if CONTAINER_REF in f.parents:
            files.append(f)
    if STAGE.exists():
        shutil.rmtree(STAGE)
    for f in files:
        dst = STAGE / f.relative_to(CONTAINER_REF)
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copyfile(f, dst)
    # package directories must exist as directories for the hook's PathFinder; their __init__ files are never executed
    total = sum(f.stat().st_size for f in files)
    print(f"staged {len(files)} reference files ({total / 1e6:.1f} MB) under {STAGE}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", action="store_true")
    if ap.parse_args().stage:
        stage()
