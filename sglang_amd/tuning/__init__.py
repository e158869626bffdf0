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
STATUS = "not asked"       # what happened, in words: the log and the reference-stack reports (`plugin_counts.gemm_selections`) carry it


def load_gemm_selections() -> bool:
    """Enable TunableOp in lookup-only mode with the committed selections. Idempotent; returns whether
    the selections are active.  SGLANG_AMD_TUNABLEOP=0 turns it off."""
    global _loaded
    if _loaded:
        return True
    import torch

    if os.environ.get("SGLANG_AMD_TUNABLEOP", "1") == "0":
        return _report("off (SGLANG_AMD_TUNABLEOP=0): library default selections")
    if not torch.cuda.is_available() or not RESULTS.exists():
        return _report("off (no GPU / no committed selections)")
    if os.environ.get("PYTORCH_TUNABLEOP_TUNING") == "1":
        return _report("off (a TunableOp tuning run is in control)")      # benchmarks/tune_gemms.py
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
            return _report(f"REJECTED by TunableOp's validators ({RESULTS.name} does not match this PyTorch / ROCm / GPU): library default selections")
        n = sum(1 for line in RESULTS.read_text().splitlines() if line and not line.startswith("Validator"))
        _report(f"loaded ({n} GEMM selections from {RESULTS.name}, lookup only; process-wide: every torch matmul of the listed shapes; "
                "SGLANG_AMD_TUNABLEOP=0 turns it off)")
        return True
    except Exception as e:                # an older / newer TunableOp API: keep the library defaults    # noqa: BLE001
        return _report(f"off ({type(e).__name__}: {e})")


def _report(what: str) -> bool:
    """One log line per process about the GEMM selections (a rejected file would otherwise be a silent slowdown)."""
    global STATUS
    STATUS = what
    import logging

    logging.getLogger("sglang_amd").info("library-GEMM selections: %s", what)
    return False
