"""Library-GEMM selection for the prefill-sized projections (the only GEMMs this path leaves to
hipBLASLt / rocBLAS).  PyTorch's TunableOp picks, per (M, N, K), the fastest solution either library
offers; the selections measured on MI355X for the BASELINE.json shapes are committed next to this file
and loaded read-only at start-up (no tuning at run time).  An entry that does not match the running
PyTorch / ROCm / GPU is ignored by TunableOp's own validators, and shapes without an entry keep the
library's default heuristic -- this file can only make a prefill GEMM faster, never change its result
beyond the libraries' own bf16 accumulation-order differences."""
from __future__ import annotations

import os
from pathlib import Path

RESULTS = Path(__file__).resolve().parent / "tunableop_gfx950.csv"
_loaded = False


def load_gemm_selections() -> bool:
    """Enable TunableOp in lookup-only mode with the committed selections. Idempotent; returns whether
    the selections are active.  SGLANG_AMD_TUNABLEOP=0 turns it off."""
    global _loaded
    if _loaded:
        return True
    import torch

    if os.environ.get("SGLANG_AMD_TUNABLEOP", "1") == "0" or not torch.cuda.is_available() or not RESULTS.exists():
        return False
    if os.environ.get("PYTORCH_TUNABLEOP_TUNING") == "1":
        return False                      # a tuning run (benchmarks/tune_gemms.py) is in control
    try:
        import tempfile

        tun = torch.cuda.tunable
        tun.enable(True)
        tun.tuning_enable(False)
        tun.record_untuned_enable(False)
        # TunableOp rewrites "its" file at exit: point that at a scratch path, never at the committed file
        tun.set_filename(os.path.join(tempfile.gettempdir(), f"sglang_amd_tunableop_{os.getpid()}.csv"))
        ok = tun.read_file(str(RESULTS))
        _loaded = bool(ok)
        if not ok:
            tun.enable(False)
        return _loaded
    except Exception:                     # an older / newer TunableOp API: keep the library defaults
        return False
