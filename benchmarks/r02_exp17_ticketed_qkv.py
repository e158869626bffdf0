"""qkv + rope + KV store of the Llama-3-8B decode layer (M = 64): GEMM + combine pair vs the one-launch ticketed form
(the last workgroup of a column block to hand in its partials finishes its heads), weights nobody touched since they
were evicted (24 rotating copies), hipGraph-timed."""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sglang_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
COPIES = 24
Hq, Hkv, D, Kd, M = 32, 8, 128, 4096, 64
N = (Hq + 2 * Hkv) * D


def graph_time(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ws = [torch.randn((N, Kd), device=dev).to(BF) * 0.02 for _ in range(COPIES)]
x = K.blocked_activation(M, Kd, dev)
x.copy_(torch.randn(x.shape, device=dev).to(BF))
cache = torch.randn((2048, D), device=dev).to(BF)
pos = torch.randint(0, 2048, (M,), device=dev)
loc = torch.randperm(4096, device=dev)[:M] + 1
kc = torch.zeros((8192, Hkv, D), dtype=BF, device=dev)
vc = torch.zeros_like(kc)
out = {}
for name, kw in (("pair_auto", dict(ticketed=False)), ("pair_8x4", dict(ticketed=False, waves_per_group=8, splits=4)),
                 ("ticket_8x3", dict(ticketed=True, waves_per_group=8, splits=3)),
                 ("ticket_8x4", dict(ticketed=True, waves_per_group=8, splits=4)),
                 ("ticket_8x5", dict(ticketed=True, waves_per_group=8, splits=5)),
                 ("pair_auto_again", dict(ticketed=False))):
    t = graph_time(lambda: [K.wstream_qkv_rope(x, w, None, pos, cache, Hq, Hkv, D, kc, vc, loc, **kw) for w in ws]) / COPIES
    out[name] = t
    print(name, round(t, 2), "us")
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp17_ticketed_qkv.json").write_text(json.dumps(out, indent=1))
