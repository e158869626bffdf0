"""Build libsglang_amd.so (gfx950 only) with hipcc, in-tree.

`python -m sglang_amd.build` cross-compiles every translation unit under
csrc/ for gfx950 (no GPU needed) and links them into sglang_amd/lib/.  The
.so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB_PATH = LIB_DIR / "libsglang_amd.so"
INCLUDE = PKG / "include"          # the packaged header (a link to the repo's include/ in the source tree)
ARCH = "gfx950"

HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-Wall",
    "-Wno-unused-function",
    "-Wno-unused-result",
    "-Wno-unused-variable",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build the gfx950 kernels)")


def sources() -> list[Path]:
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _deps_mtime() -> float:
    hdrs = list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.inc")) + list(INCLUDE.glob("*.h")) + list((PKG.parent / "include").glob("*.h"))
    return max([h.stat().st_mtime for h in hdrs if h.exists()] + [0.0])


def _compile_one(src: Path, obj: Path, verbose: bool) -> None:
    # the C-ABI header: packaged copy first (installed tree), the repo's include/ as the source-tree fallback
    incs = [f"-I{d}" for d in (INCLUDE, PKG.parent / "include") if (d / "sglang_amd.h").exists()]
    cmd = [_hipcc(), *HIPCC_FLAGS, *incs, "-c", str(src), "-o", str(obj)]
    if src.suffix == ".cpp":
        cmd.insert(1, "-x")
        cmd.insert(2, "hip")
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile (if stale) and return the path of libsglang_amd.so."""
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = LIB_DIR / "obj"
    obj_dir.mkdir(exist_ok=True)
    srcs = sources()
    dep_m = _deps_mtime()
    jobs = []
    for s in srcs:
        o = obj_dir / (s.name + ".o")
        stale = force or not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, dep_m)
        if stale:
            jobs.append((s, o))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(lambda so: _compile_one(so[0], so[1], verbose), jobs))
    objs = [obj_dir / (s.name + ".o") for s in srcs]
    need_link = bool(jobs) or not LIB_PATH.exists() or any(
        o.stat().st_mtime > LIB_PATH.stat().st_mtime for o in objs
    )
    if need_link:
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB_PATH)]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print(f"built {p}")
