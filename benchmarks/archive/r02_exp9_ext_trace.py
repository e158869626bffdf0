"""Where a KV tile's time goes in the 8-wave extend attention kernel (variant library built with -DEXT_TRACE):
shader clocks per phase, averaged over waves, for the cold (4 x 1024, no prefix) and warm (60 x 128 over 896) shapes."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ["SGLANG_AMD_LIB"] = str(ROOT / "benchmarks" / "variants" / "lib_ext_trace.so")
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from sglang_amd import kernels as K, native  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
Hq, Hkv, D = 32, 8, 128
slots = 64 * 1200 + 4096
kc = torch.randn((slots, Hkv, D), device=dev).to(BF)
vc = torch.randn((slots, Hkv, D), device=dev).to(BF)
ctx = 1160
r2t = torch.zeros((65, 4100), dtype=torch.int32, device=dev)
perm = (torch.randperm(slots - 1, device=dev) + 1).to(torch.int32)
for b in range(64):
    r2t[b + 1, :1088] = perm[b * 1088:(b + 1) * 1088]
for b in range(2):
    r2t[b + 1, :4096] = perm[b * 4096:(b + 1) * 4096]
if os.environ.get("EXT_LOCAL"):            # every row from 512 slots: the gathers hit L2 (is the staging time latency?)
    r2t = (r2t % 512 + 1).to(torch.int32)
fn = native.lib().sgl_amd_debug_ext_trace
fn.argtypes = [C.c_void_p]
fn.restype = C.c_int
out = {}
FLAGS = int(os.environ.get("EXT_FLAGS", "0"))          # 0: the 32x32 two-score-set kernel; 2: the ping-pong kernel
native.call("sgl_amd_debug_extend_attention_shape", 0, FLAGS)
names = {2: ["issue", "s_t", "softmax", "pv", "commit", "barrier"], 0: ["commit", "s_exp", "mask", "pv_max", "decision", "barrier"]}[FLAGS]
for name, nreq, pre, e in (("cold", 4, 0, 1024), ("warm", 60, 896, 128), ("long", 2, 0, 4096)):
    T = nreq * e
    qx = torch.randn((T, Hq, D), device=dev).to(BF)
    ox = torch.empty_like(qx)
    seq_x = torch.full((nreq,), pre + e, dtype=torch.int32, device=dev)
    pre_x = torch.full((nreq,), pre, dtype=torch.int32, device=dev)
    qo = (torch.arange(nreq + 1, device=dev) * e).to(torch.int32)
    pool_x = torch.arange(1, nreq + 1, device=dev)
    tiles = (e * 4 + 255) // 256
    trace = torch.zeros((nreq * Hkv * tiles * 64,), dtype=torch.int64, device=dev)
    assert fn(trace.data_ptr()) == 0
    for _ in range(3):
        K.extend_attention(qx, ox, kc, vc, r2t, pool_x, seq_x, pre_x, qo, e, D ** -0.5, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.extend_attention(qx, ox, kc, vc, r2t, pool_x, seq_x, pre_x, qo, e, D ** -0.5, True)
    e1.record()
    torch.cuda.synchronize()
    t = trace.cpu().view(-1, 8, 8).double()           # [wg][wave][field]
    n_t = t[:, :, 6]
    tot = t[:, :, :6].sum(dim=(0, 1)) / n_t.sum()     # clocks per tile per wave
    r = {nm: round(float(v), 1) for nm, v in zip(names, tot)}
    r["per_tile_total"] = round(float(tot.sum()), 1)
    r["event_us"] = e0.elapsed_time(e1) * 1e3
    r["tile_iters"] = float(n_t[:, 0].sum())
    # spread of workgroup start times (100 MHz? shader clock units as returned) to see the rounds
    st = t[:, 0, 7]
    r["start_spread_clk"] = float(st.max() - st.min())
    out[name] = r
    print(name, json.dumps(r))
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / (("r03_exp9_ext_trace_pingpong" if FLAGS == 2 else "r03_exp9_ext_trace_32x32") + ("_local" if os.environ.get("EXT_LOCAL") else "") + ".json")).write_text(json.dumps(out, indent=1))
