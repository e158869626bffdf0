"""Is the gate_up launch slower on the model's own weight tensors than on fresh copies of them?  (bench.py's roofline leg
times 44 us on the former, benchmarks/r02_exp18 40 us on fresh tensors.)"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sglang_amd import kernels as K  # noqa: E402
from sglang_amd.harness.engine import ModelRunner  # noqa: E402
from sglang_amd.harness.models import CONFIGS  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
cfg = CONFIGS["llama-3-8b"]
runner = ModelRunner(cfg, max_total_tokens=64 * 1152 + 4096, max_running_requests=64, max_context_len=1160, page_size=1, device=dev,
                     use_graph=False)
mlps = [layer.mlp for layer in runner.model.layers]
M, H = 64, cfg.hidden_size


def graph_time(fn, reps=5):
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


xb = K.blocked_activation(M, H, dev)
xb.copy_(torch.randn(xb.shape, device=dev).to(BF))
own = [m.gate_up_proj.weight.data for m in mlps]
out = {"own_ptr_mod_2MiB": [int(w.data_ptr() % (2 << 20)) for w in own[:4]], "own_std": float(own[0].float().std()),
       "own_stride": list(own[0].stride())}
out["own"] = graph_time(lambda: [K.wstream_gemm(xb, w, epilogue="silu_and_mul", out_blocked=True) for w in own]) / len(own)
clones = [w.clone() for w in own]
out["clone_ptr_mod_2MiB"] = [int(w.data_ptr() % (2 << 20)) for w in clones[:4]]
out["clones"] = graph_time(lambda: [K.wstream_gemm(xb, w, epilogue="silu_and_mul", out_blocked=True) for w in clones]) / len(own)
rnd = [torch.randn_like(w, dtype=torch.float32).mul_(0.02).to(BF) for w in own[:16]]
out["random_0.02"] = graph_time(lambda: [K.wstream_gemm(xb, w, epilogue="silu_and_mul", out_blocked=True) for w in rnd]) / len(rnd)
out["own_again"] = graph_time(lambda: [K.wstream_gemm(xb, w, epilogue="silu_and_mul", out_blocked=True) for w in own]) / len(own)
for k, v in out.items():
    print(k, v)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_exp19_model_weights.json").write_text(json.dumps(out, indent=1))
