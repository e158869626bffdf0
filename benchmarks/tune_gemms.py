"""Tune the library GEMMs of the bench workload with PyTorch TunableOp on one MI355X and write
sglang_amd/tuning/tunableop_gfx950.csv (committed).  Shapes: the prefill projections of
`--model` at TP = 1, 2, 4, 8 (cold: G*1024 tokens, warm: (B-G)*128 tokens, B = 64 * TP) and the decode
projections that are left to the library (batches above the weight-streaming GEMM's row limit).

    python benchmarks/tune_gemms.py --out gpurun_out/tunableop_gfx950.csv
"""
import argparse
import os
import sys
from pathlib import Path

os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "40")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "5")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd import kernels as K  # noqa: E402
from sglang_amd.harness.models import CONFIGS  # noqa: E402


def shapes(cfg, tps, groups=4, per_group=16, prefix=896, unique=128):
    out = set()
    H, D, I, V = cfg.hidden_size, cfg.head_dim, cfg.intermediate_size, cfg.vocab_size
    for tp in tps:
        G, B = groups * tp, groups * per_group * tp
        qkv = (cfg.num_attention_heads // tp + 2 * max(1, cfg.num_key_value_heads // tp)) * D
        proj = [(qkv, H), (H, cfg.num_attention_heads // tp * D), (2 * I // tp, H), (H, I // tp)]
        for M in (G * (prefix + unique), (B - G) * unique):
            for n, k in proj:
                out.add((M, n, k))
        for M in (G, B - G, B):                      # lm_head rows (last token of every request) and decode
            for n, k in proj + [(V // tp, H)]:
                if not K.wstream_preferred(M, n, k):
                    out.add((M, n, k))
    return sorted(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--tp", default="1,2,4,8")
    ap.add_argument("--out", default="gpurun_out/tunableop_gfx950.csv")
    a = ap.parse_args()
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    torch.cuda.tunable.set_filename(a.out)
    dev = torch.device("cuda:0")
    todo = shapes(CONFIGS[a.model], [int(t) for t in a.tp.split(",")])
    print(f"{len(todo)} GEMM shapes")
    for M, n, k in todo:
        x = torch.randn((M, k), device=dev).to(torch.bfloat16)
        w = (torch.randn((n, k), device=dev) * 0.02).to(torch.bfloat16)
        F.linear(x, w)
        torch.cuda.synchronize()
        print("tuned", M, n, k, flush=True)
    print(f"{len(torch.cuda.tunable.get_results())} selections; TunableOp writes {a.out} at exit")


if __name__ == "__main__":
    main()
