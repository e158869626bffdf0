"""ctypes binding of libsglang_amd.so (the C ABI declared in include/sglang_amd.h).

The library is the product: there is no Python / torch fallback for any op.
If the shared object is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
from pathlib import Path
from typing import Dict, List

_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_I32 = ctypes.c_int
_F32 = ctypes.c_float

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "lib" / "libsglang_amd.so"
# the header ships inside the package (sglang_amd/include/, listed in package-data) so that an installed,
# non-editable tree can bind and rebuild; in the source tree that entry is a link to the repo's include/
HEADER_PATH = next((h for h in (PKG / "include" / "sglang_amd.h", PKG.parent / "include" / "sglang_amd.h") if h.exists()),
                   PKG / "include" / "sglang_amd.h")

_lib = None
_sigs: Dict[str, List] = {}


def _parse_header() -> Dict[str, List]:
    """Derive argtypes from include/sglang_amd.h so the binding cannot drift."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    sigs: Dict[str, List] = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(sgl_amd_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(_P)
                elif a.startswith("int64_t"):
                    argtypes.append(_I64)
                elif a.startswith("float"):
                    argtypes.append(_F32)
                elif a.startswith("int"):
                    argtypes.append(_I32)
                else:
                    raise RuntimeError(f"sglang_amd.h: cannot map parameter '{a}' of {name}")
        sigs[name] = [ret, argtypes]
    return sigs


def declared_symbols() -> List[str]:
    return sorted(_parse_header().keys())


def lib() -> ctypes.CDLL:
    global _lib, _sigs
    if _lib is not None:
        return _lib
    # ORDER MATTERS: torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  It must be
    # in the process before our library is dlopen'ed so that our NEEDED libamdhip64.so.7 resolves
    # to the SAME runtime instance torch uses (streams, graphs and device pointers are shared).
    # Loading us first would pull a second HIP runtime from /opt/rocm and every launch would fail
    # with hipErrorNoDevice.
    import torch  # noqa: F401

    path = Path(os.environ.get("SGLANG_AMD_LIB", str(LIB_PATH)))
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: the gfx950 HIP extension is not built. "
            "Run `python -m sglang_amd.build` (or __graft_entry__.build()). "
            "There is no CPU/torch fallback for this path."
        )
    cdll = ctypes.CDLL(str(path))
    _sigs = _parse_header()
    for name, (ret, argtypes) in _sigs.items():
        try:
            fn = getattr(cdll, name)
        except AttributeError as e:
            raise RuntimeError(f"{path} does not export {name} (declared in sglang_amd.h)") from e
        fn.argtypes = argtypes
        fn.restype = {"int": ctypes.c_int, "int64_t": ctypes.c_int64}.get(ret, ctypes.c_char_p)
    _lib = cdll
    return cdll


def last_error() -> str:
    msg = lib().sgl_amd_last_error()
    return msg.decode() if msg else ""


def call(name: str, *args) -> None:
    """Call an int-returning entry point; raise RuntimeError on a non-zero status."""
    fn = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (status {rc}): {last_error()}")
