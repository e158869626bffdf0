"""Would the start of a weight-streaming GEMM go faster if the bytes it asks for first were already in the memory-side
cache?  Per projection (rotating cold weights, M = 64, chunk-major x): [filler] -> GEMM vs [prefetch of the first PD chunks
of every row and K range] -> GEMM, the prefetch kernel timed on its own too (variant library built with -DWS_EXPERIMENT)."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ["SGLANG_AMD_LIB"] = str(ROOT / "benchmarks" / "variants" / "lib_ws_exp.so")
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from sglang_amd import kernels as K, native  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
COPIES = 16
lib = native.lib()
fn = lib.sgl_amd_debug_prefetch_head
fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
fn.restype = C.c_int
sink = torch.zeros(4, dtype=torch.int32, device=dev)


def graph_time(f, reps=10):
    f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out = {}
for name, N, Kd, ep in (("qkv", 6144, 4096, "none"), ("o_proj", 4096, 4096, "none"), ("gate_up", 28672, 4096, "silu_and_mul"),
                        ("down", 4096, 14336, "none")):
    ws = [torch.randn((N, Kd), device=dev).to(BF) * 0.02 for _ in range(COPIES)]
    x = torch.randn((64, Kd), device=dev).to(BF)
    xb = x.view(64, Kd // 128, 128).permute(1, 0, 2).contiguous()
    nw, tpw, s = K.choose_wstream_decomposition(64, N, Kd, False, ep == "silu_and_mul")
    pd = 4 if tpw == 1 else 3
    st = torch.cuda.current_stream

    def pre(w):
        fn(w.data_ptr(), w.stride(0), N, Kd, s, pd, sink.data_ptr(), st().cuda_stream)

    def gemm(w):
        K.wstream_gemm(xb, w, epilogue=ep)

    t_gemm = graph_time(lambda: [gemm(w) for w in ws]) / COPIES
    t_pre = graph_time(lambda: [pre(w) for w in ws]) / COPIES
    t_both = graph_time(lambda: [(pre(w), gemm(w)) for w in ws]) / COPIES
    t_other = graph_time(lambda: [(pre(ws[(i + COPIES // 2) % COPIES]), gemm(w)) for i, w in enumerate(ws)]) / COPIES
    mb = N * s * pd * 256 / 1e6
    out[name] = {"decomposition": [nw, tpw, s], "prefetched_MB": mb, "gemm_us": t_gemm, "prefetch_us": t_pre, "prefetch_then_gemm_us": t_both,
                 "prefetch_other_then_gemm_us": t_other, "gemm_after_own_prefetch_us": t_both - t_pre,
                 "gemm_after_other_prefetch_us": t_other - t_pre}
    print(name, json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out[name].items()}))
    del ws
    torch.cuda.empty_cache()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp16_prefetch.json").write_text(json.dumps(out, indent=1))
