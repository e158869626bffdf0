"""Sampler.forward at the decode shape ([64, 128256] bf16 logits, temperature 1, top-k 50, top-p 0.9, seeded), its parts,
and the greedy arg-max beside it (hipGraph-timed, 8 calls per graph)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402
from sglang_amd.layers.sampler import LogitsProcessorOutput, Sampler, SamplingBatchInfo  # noqa: E402

DEV = torch.device("cuda:0")


def graph_time(fn, n=8, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


def main():
    B, V = 64, 128256
    lg = (torch.randn((B, V), device=DEV) * 2.0).to(torch.bfloat16)
    info = SamplingBatchInfo(torch.ones((B, 1), device=DEV), torch.full((B,), 0.9, device=DEV),
                             torch.full((B,), 50, dtype=torch.int32, device=DEV), torch.zeros(B, device=DEV), False,
                             need_top_p_sampling=True, need_top_k_sampling=True,
                             sampling_seed=torch.arange(B, device=DEV, dtype=torch.int64) + 1234)
    pos = torch.arange(B, device=DEV, dtype=torch.int64) + 1000
    smp = Sampler()
    out = {"sampler_us": graph_time(lambda: smp(LogitsProcessorOutput(next_token_logits=lg), info, positions=pos))}
    f32 = lg.float()
    probs = K.softmax_temperature_(f32.clone(), info.temperatures)
    out["widen_us"] = graph_time(lambda: lg.float())
    out["softmax_us"] = graph_time(lambda: K.softmax_temperature_(f32, info.temperatures))   # (in place on garbage from the 2nd call on: same work)
    out["sample_us"] = graph_time(lambda: K.top_k_top_p_min_p_sample(probs, info.top_ks, info.top_ps, None, info.sampling_seed, pos))
    out["argmax_us"] = graph_time(lambda: K.argmax(lg))
    # temperature-only sampling (top_k = all, top_p = 1: sampler.py's "simple sampling case"): every token's gumbel score
    info_s = SamplingBatchInfo(torch.ones((B, 1), device=DEV), torch.ones(B, device=DEV), torch.full((B,), 1 << 30, dtype=torch.int32, device=DEV),
                               torch.zeros(B, device=DEV), False, sampling_seed=torch.arange(B, device=DEV, dtype=torch.int64) + 1234)
    out["sampler_temperature_only_us"] = graph_time(lambda: smp(LogitsProcessorOutput(next_token_logits=lg), info_s, positions=pos))
    out["sample_unfiltered_us"] = graph_time(lambda: K.top_k_top_p_min_p_sample(probs, None, None, None, info.sampling_seed, pos, filtered=False))
    print(json.dumps(out))
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/r03_exp6_sampler.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
