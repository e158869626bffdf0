"""Round 6: the lm_head weight stream [64 x 128256 x 4096] over (waves per group, tiles per wave)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def gt(fn, launches, reps=8, warm=3):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(warm):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * launches) * 1e3


N, Kd, M = 128256, 4096, 64
ws = [(torch.randn((N, Kd), device=dev) * 0.02).to(BF) for _ in range(2)]      # 2 x 1.05 GB: nothing stays in the 256 MiB cache
x = K.blocked_activation(M, Kd, dev); x.copy_(torch.randn(x.shape, device=dev).to(BF))
out = {}
for nw, tpw in ((4, 2), (3, 2), (2, 2), (8, 1), (7, 1), (6, 1), (5, 1), (4, 1)):
    try:
        t = gt(lambda: [K.wstream_gemm(x, w, waves_per_group=nw, tiles_per_wave=tpw, splits=1) for w in ws], len(ws))
    except Exception as e:  # noqa: BLE001
        t = None
        print(nw, tpw, "failed", str(e)[:80])
    out[f"nw{nw}_tpw{tpw}"] = t
    print(f"nw={nw} tiles/wave={tpw}: {t}", flush=True)
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps({"what": "lm_head [64, 128256, 4096] weight stream, us per launch", "us": out}, indent=1))
