"""Timeline of one cascade chunk launch on the bench batch (variant library built with -DCASC_TRACE)."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ["SGLANG_AMD_LIB"] = str(ROOT / "benchmarks" / "variants" / "lib_casc_trace.so")
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from sglang_amd import kernels as K, native  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
Hq, Hkv, D, G, P, prefix, len_k, ctx = 32, 8, 128, 4, 16, 896, 1088, 1160
B = G * P
slots = B * 1200 + 4096
kc = torch.randn((slots, Hkv, D), device=dev).to(BF)
vc = torch.randn((slots, Hkv, D), device=dev).to(BF)
r2t = torch.zeros((B + 1, ctx), dtype=torch.int32, device=dev)
perm = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
off = 0
for b in range(B):
    r2t[b + 1, :len_k] = perm[off: off + len_k]
    off += len_k
    r2t[b + 1, :prefix] = r2t[(b // P) * P + 1, :prefix]
pool = torch.arange(1, B + 1, device=dev)
seq = torch.full((B,), len_k, dtype=torch.int32, device=dev)
q = torch.randn((B, Hq, D), device=dev).to(BF)
o = torch.empty_like(q)
ws = K.CascadeWorkspace(B, Hq, D, ctx, dev)
K.cascade_plan(ws, r2t, pool, seq, Hq, Hkv)
items = int(ws.plan[0])
fn = native.lib().sgl_amd_debug_casc_trace
fn.argtypes = [C.c_void_p]
fn.restype = C.c_int
trace = torch.zeros((8192, 8), dtype=torch.int64, device=dev)
assert fn(trace.data_ptr()) == 0
for it in range(4):
    for _ in range(20):
        K.cascade_decode_attention(ws, q, kc, vc, o, r2t, pool, seq, D ** -0.5)
    torch.cuda.synchronize()
    trace.zero_()
    K.cascade_decode_attention(ws, q, kc, vc, o, r2t, pool, seq, D ** -0.5)
    torch.cuda.synchronize()
t = trace.cpu()[: items * Hkv].double() * 0.01
base = t[:, 0].min()
t = t - base
names = ["entry", "record", "k_staged", "scores", "v_staged", "stored"]
shared = 28 * Hkv


def stats(rows, label):
    r = {}
    for i, nm in enumerate(names):
        col = rows[:, i]
        r[nm] = [round(float(col.min()), 2), round(float(col.median()), 2), round(float(col.max()), 2)]
    r["dur_p50"] = round(float((rows[:, 5] - rows[:, 0]).median()), 2)
    r["dur_max"] = round(float((rows[:, 5] - rows[:, 0]).max()), 2)
    print(label, json.dumps(r))
    return r


out = {"items": items, "units": items * Hkv, "shared": stats(t[:shared], "shared"), "private": stats(t[shared:], "private"),
       "late_starters": int((t[:, 0] > 2.0).sum()), "entry_sorted_tail": [round(float(v), 2) for v in t[:, 0].sort().values[-8:]]}
print(json.dumps({k: v for k, v in out.items() if k not in ("shared", "private")}))
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp8_casc_trace.json").write_text(json.dumps(out, indent=1))
