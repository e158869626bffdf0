"""Cascade chunk-kernel variants (images per item x workgroups per CU), each from its own build of the library
(SGLANG_AMD_LIB), on the bench batch; plus the cost of the workgroups behind the end of the item list."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
from sglang_amd import kernels as K
dev = torch.device("cuda:0"); BF = torch.bfloat16
def graph_time(fn, reps=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
Hq, Hkv, D, G, P, prefix, len_k, ctx = 32, 8, 128, 4, 16, 896, 1088, 1160
B = G * P; slots = B * 1200 + 4096
kc = torch.randn((slots, Hkv, D), device=dev).to(BF); vc = torch.randn((slots, Hkv, D), device=dev).to(BF)
r2t = torch.zeros((B + 1, ctx), dtype=torch.int32, device=dev)
perm = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
off = 0
for b in range(B):
    r2t[b + 1, :len_k] = perm[off: off + len_k]; off += len_k
    r2t[b + 1, :prefix] = r2t[(b // P) * P + 1, :prefix]
pool = torch.arange(1, B + 1, device=dev); seq = torch.full((B,), len_k, dtype=torch.int32, device=dev)
q = torch.randn((B, Hq, D), device=dev).to(BF); o = torch.empty_like(q); o2 = torch.empty_like(q)
ws = K.CascadeWorkspace(B, Hq, D, ctx, dev)
K.cascade_plan(ws, r2t, pool, seq, Hq, Hkv)
items = int(ws.plan[0])
res = {"items": items}
res["us_bound_grid"] = graph_time(lambda: K.cascade_decode_attention(ws, q, kc, vc, o, r2t, pool, seq, D ** -0.5))
os.environ["SGL_AMD_CASCADE_UNITS"] = str(items)
res["us_exact_grid"] = graph_time(lambda: K.cascade_decode_attention(ws, q, kc, vc, o, r2t, pool, seq, D ** -0.5))
K.decode_attention(q, kc, vc, o2, r2t, pool, seq, D ** -0.5)
res["err"] = float((o.float() - o2.float()).abs().max())
print("RESULT " + json.dumps(res))
''' % str(ROOT)
out = {}
for v in sys.argv[1:]:
    env = dict(os.environ, SGLANG_AMD_LIB=str(ROOT / "benchmarks" / "variants" / f"lib_casc_{v}.so"))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    out[v] = json.loads(line[0][7:]) if line else {"error": r.stderr[-500:]}
    print(v, out[v])
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp2.json").write_text(json.dumps(out, indent=1))
