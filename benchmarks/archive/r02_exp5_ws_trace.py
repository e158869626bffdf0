"""Timeline of one weight-streaming GEMM launch (variant library built with -DWS_TRACE): per workgroup, the chip-wide
100 MHz clock at entry, first chunk landed, last chunk multiplied, epilogue stored."""
import ctypes
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ["SGLANG_AMD_LIB"] = str(ROOT / "benchmarks" / "variants" / "lib_ws_trace.so")
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from sglang_amd import kernels as K, native  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
lib = ctypes.CDLL(os.environ["SGLANG_AMD_LIB"])
native.lib()
out = {}
for name, N, Kd, ep in (("qkv", 6144, 4096, "none"), ("o_proj", 4096, 4096, "none"), ("gate_up", 28672, 4096, "silu_and_mul"),
                        ("down", 4096, 14336, "none")):
    ws = [torch.randn((N, Kd), device=dev).to(BF) * 0.02 for _ in range(8)]
    x = torch.randn((64, Kd), device=dev).to(BF)
    nw, s = K.choose_wstream_config(64, N, Kd, ep != "none" and False, ep == "silu_and_mul")
    trace = torch.zeros((4096, 4), dtype=torch.int64, device=dev)
    native_lib = native.lib()
    import ctypes as C
    fn = getattr(native_lib, "sgl_amd_debug_ws_trace", None) or lib.sgl_amd_debug_ws_trace
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    assert fn(trace.data_ptr()) == 0
    rows = []
    for it in range(6):
        trace.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for j in range(24):                                   # predecessors keep the queue and the clocks warm
            K.wstream_gemm(x, ws[(it + j + 1) % 8], epilogue=ep)
        e0.record()
        K.wstream_gemm(x, ws[it % 8], epilogue=ep)
        e1.record()
        torch.cuda.synchronize()
        t = trace.cpu()
        t = t[t[:, 0] > 0].double() * 0.01                     # us
        base = t[:, 0].min()
        t = t - base
        rows.append({"wgs": int(t.shape[0]), "entry_max": float(t[:, 0].max()), "entry_p50": float(t[:, 0].median()),
                     "first_landed_min": float(t[:, 1].min()), "first_landed_p50": float(t[:, 1].median()),
                     "first_landed_max": float(t[:, 1].max()),
                     "last_mult_min": float(t[:, 2].min()), "last_mult_p50": float(t[:, 2].median()),
                     "last_mult_max": float(t[:, 2].max()), "stored_max": float(t[:, 3].max()),
                     "stream_p50": float((t[:, 2] - t[:, 1]).median()), "event_us": e0.elapsed_time(e1) * 1e3})
    r = rows[-1]
    r.update(MB=N * Kd * 2 / 1e6, nw=nw, splits=s)
    out[name] = r
    print(name, json.dumps({k: round(v, 2) if isinstance(v, float) else v for k, v in r.items()}))
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp5_ws_trace.json").write_text(json.dumps(out, indent=1))
