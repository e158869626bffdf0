"""Merge the rocprofv3 --pmc passes of benchmarks/pmc_workload.py (one output directory per counter set) into
profiles/rNN_pmc.json: per phase (split at the mfma_probe_kernel sentinels) and per kernel, the average counter
values per dispatch, plus the phase totals bench.py quotes (HBM bytes per decode step, MFMA-busy fraction).

    python benchmarks/summarize_pmc_phases.py OUT.json MODEL BATCH DECODE_STEPS DIR [DIR ...]
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict
from pathlib import Path

PHASES = ["warmup", "prefill_cold", "prefill_warm", "decode", "tail"]
CSRC = Path(__file__).resolve().parent.parent / "sglang_amd" / "csrc"


def source_stamp():
    """What the counters were collected ON: sha256 of every kernel source (bench.py quotes a kernel's counters only
    while the sources that kernel is built from still hash to this) + the revision when the caller exports it (the GPU
    box has no .git)."""
    return {"git_revision": os.environ.get("SGLANG_AMD_GIT_REV") or None,
            "sources": {f.name: hashlib.sha256(f.read_bytes()).hexdigest()[:16] for f in sorted(CSRC.iterdir()) if f.is_file()}}


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:160]


def main(out_path, model, batch, decode_steps, dirs):
    phases = {p: {"kernels": defaultdict(lambda: defaultdict(lambda: [0, 0.0])), "totals": defaultdict(float)} for p in PHASES}
    for d in dirs:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            rows = list(csv.DictReader(open(f)))
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            # phase of every dispatch id: advance at each sentinel dispatch
            phase_of, ph, last = {}, 0, None
            for r in rows:
                did = int(r["Dispatch_Id"])
                if did == last:
                    continue
                last = did
                if "mfma_probe_kernel" in r["Kernel_Name"]:
                    ph = min(ph + 1, len(PHASES) - 1)
                    phase_of[did] = None
                else:
                    phase_of[did] = PHASES[ph]
            for r in rows:
                p = phase_of[int(r["Dispatch_Id"])]
                if p is None:
                    continue
                # one template instance can serve several operators (gate_up and lm_head both run the two-tile GEMM):
                # key by launch size too so that per-dispatch averages stay per operator
                key = short(r["Kernel_Name"]) + " grid=" + str(r.get("Grid_Size", "?"))
                a = phases[p]["kernels"][key][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                phases[p]["totals"][r["Counter_Name"]] += float(r["Counter_Value"])
    out = {"model": model, "batch": int(batch), "decode_steps": int(decode_steps), "stamp": source_stamp(),
           "source": "rocprofv3 --pmc passes of benchmarks/pmc_workload.py (separate pass per counter set; eager launches)",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (gfx950: double FETCH_SIZE for wide streaming reads); "
                    "SQ_* / GRBM_* raw counts per dispatch", "phases": {}}
    for p in PHASES:
        if p in ("warmup", "tail"):
            continue
        ks = {}
        for name, ctrs in phases[p]["kernels"].items():
            rec = {"dispatches": max(n for n, _ in ctrs.values())}
            for c, (n, tot) in ctrs.items():
                rec[c] = tot / n
            ks[name] = rec
        tot = dict(phases[p]["totals"])
        ph = {"totals": tot, "kernels": dict(sorted(ks.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0) * kv[1]["dispatches"]))}
        if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
            ph["hbm_bytes"] = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024
            if p == "decode":
                ph["hbm_bytes_per_step"] = ph["hbm_bytes"] / int(decode_steps)
        if tot.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs on gfx950: busy SIMD-cycles / (active cycles per XCD x 1024 SIMDs)
            ph["mfma_util"] = tot["SQ_VALU_MFMA_BUSY_CYCLES"] * 8 / (tot["GRBM_GUI_ACTIVE"] * 1024)
        out["phases"][p] = ph
    json.dump(out, open(out_path, "w"), indent=1)
    for p, ph in out["phases"].items():
        print(p, {k: v for k, v in ph.items() if k not in ("kernels", "totals")})
        for name, rec in list(ph["kernels"].items())[:12]:
            print("   ", rec["dispatches"], name[:90], {k: round(v, 1) for k, v in rec.items() if k != "dispatches"})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
