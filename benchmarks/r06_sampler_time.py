"""Round 6: the sampler at the decode shape ([64, 128256] bf16 logits, temperature 1, top-k 50, top-p 0.9, seeded): Sampler.forward end
to end and its pieces, the filtered kernel as column ranges against one workgroup per row."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402
from sglang_amd.layers.sampler import LogitsProcessorOutput, Sampler, SamplingBatchInfo  # noqa: E402

dev = torch.device("cuda:0")


def graph_time(fn, reps=20, warm=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(warm):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


B, V = 64, 128256
lg = (torch.randn((B, V), device=dev) * 2.0).to(torch.bfloat16)
info = SamplingBatchInfo(torch.ones((B, 1), device=dev), torch.full((B,), 0.9, device=dev), torch.full((B,), 50, dtype=torch.int32, device=dev),
                         torch.zeros(B, device=dev), False, need_top_p_sampling=True, need_top_k_sampling=True,
                         sampling_seed=torch.arange(B, device=dev, dtype=torch.int64) + 1234)
pos = torch.full((B,), 1024, dtype=torch.int64, device=dev)
smp = Sampler()
out = {}
out["sampler_forward_us"] = graph_time(lambda: smp(LogitsProcessorOutput(next_token_logits=lg), info, positions=pos))
out["widen_us"] = graph_time(lambda: lg.float())
f32 = lg.float()
out["softmax_us"] = graph_time(lambda: K.softmax_temperature_(f32.clone(), info.temperatures)) - graph_time(lambda: f32.clone())
probs = K.softmax_temperature_(lg.float(), info.temperatures)
for r in (0, 2, 4, 8, 16):
    out[f"sample_ranges_{r}_us"] = graph_time(lambda: K.top_k_top_p_min_p_sample(probs, info.top_ks, info.top_ps, None, info.sampling_seed, pos, ranges=r))
out["softmax_from_bf16_us"] = graph_time(lambda: K.softmax_temperature_from_bf16(lg, info.temperatures))
out["sample_from_bf16_logits_us"] = graph_time(lambda: K.sample_from_bf16_logits(lg, info.temperatures, info.top_ks, info.top_ps, None, info.sampling_seed, pos))
# top-p only (no top-k): every range emits its 64 largest, 1024 candidates are ranked
allk = torch.full((B,), 1 << 30, dtype=torch.int32, device=dev)
_, fb = K.sample_from_bf16_logits(lg, info.temperatures, allk, info.top_ps, None, info.sampling_seed, pos, return_fallback=True)
out["top_p_only_rows_redone_the_long_way"] = int(fb.sum())
out["sample_from_bf16_logits_top_p_only_us"] = graph_time(lambda: K.sample_from_bf16_logits(lg, info.temperatures, allk, info.top_ps, None, info.sampling_seed, pos))
out["two_calls_top_p_only_us"] = graph_time(lambda: K.top_k_top_p_min_p_sample(K.softmax_temperature_from_bf16(lg, info.temperatures), allk, info.top_ps, None, info.sampling_seed, pos))
# fp32 logits (what the reference's LogitsProcessor hands its Sampler): the same one-call path
lf = lg.float()
out["sampler_forward_fp32_logits_us"] = graph_time(lambda: smp(LogitsProcessorOutput(next_token_logits=lf), info, positions=pos))
out["softmax_then_sample_fp32_logits_us"] = graph_time(lambda: K.top_k_top_p_min_p_sample(K.softmax_temperature_(lf.clone(), info.temperatures), info.top_ks, info.top_ps, None, info.sampling_seed, pos)) - graph_time(lambda: lf.clone())
out["algorithmic_bytes"] = B * V * 4
out["frac_of_hbm_8TBps"] = B * V * 4 / out["sampler_forward_us"] / 1e6 / 8.0
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps(out, indent=1))
