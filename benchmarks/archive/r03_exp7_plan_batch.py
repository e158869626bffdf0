"""The once-per-step cascade plan kernel over batch sizes (groups of 16 requests sharing an 896-token prefix)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402
from benchmarks.cascade_micro import slot_table, timeit  # noqa: E402

DEV = torch.device("cuda:0")
out = {}
for B in (64, 128, 256, 512):
    r2t, slots = slot_table("allocator", B, 16, 1088, 896, 128, 1160)
    pool = torch.arange(1, B + 1, device=DEV)
    seq = torch.full((B,), 1088, dtype=torch.int32, device=DEV)
    ws = K.CascadeWorkspace(B, 32, 128, 1160, DEV)
    out[B] = timeit(lambda: K.cascade_plan(ws, r2t, pool, seq, 32, 8))
    summ = K.cascade_plan_summary(ws, B)
    assert summ["n_groups"] == B // 16, summ["n_groups"]
print(json.dumps(out))
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/r03_exp7_plan_batch.json").write_text(json.dumps(out))
