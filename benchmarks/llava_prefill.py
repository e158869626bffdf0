"""BASELINE configs[4]: LLaVA-1.6-7B image + text prefill on one MI355X (synthetic weights): CLIP ViT-L/14-336 over
the 5 anyres tiles of an image (2880 image tokens), projector, insert into the radix cache, language-model prefill;
then a second question about the SAME image (radix hit over the whole image: no encoder run, text-only prefill)."""
import json
import random
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sglang_amd.harness.engine import Engine, ModelRunner, Req  # noqa: E402
from sglang_amd.harness.llava import ClipVisionConfig, MultimodalItem, pad_input_ids  # noqa: E402
from sglang_amd.harness.models import CONFIGS  # noqa: E402

dev = torch.device("cuda:0")
cfg, vc = CONFIGS["llava-1.6-7b"], ClipVisionConfig()
B, tiles, text_len = 8, 5, 64
ntok = tiles * vc.image_feature_len + text_len + 8
runner = ModelRunner(cfg, max_total_tokens=2 * B * (ntok + 16) + 4096, max_running_requests=2 * B, max_context_len=ntok + 32,
                     device=dev, use_graph=False, vision_config=vc)
eng = Engine(runner)
rnd = random.Random(0)
IMG = cfg.vocab_size + 1
images = [torch.randn((tiles, 3, vc.image_size, vc.image_size)) for _ in range(B)]
heads = [[rnd.randrange(cfg.vocab_size) for _ in range(8)] for _ in range(B)]      # system text in front of each image


def make(i, rid):
    items = [MultimodalItem(images[i])]
    ids = pad_input_ids(heads[i] + [IMG] + [rnd.randrange(cfg.vocab_size) for _ in range(text_len)],
                        IMG, items, vc.image_feature_len)
    q = Req(rid, ids, 2)
    q.mm_items = items
    return q


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0


# vision tower alone
x = torch.cat(images).to(dev)
runner.vision.encode_images(x[:tiles])
t_vis = timed(lambda: runner.vision.encode_images(x))
T = vc.num_patches + 1
L = runner.vision.tower.n_layers
H, I = vc.hidden_size, vc.intermediate_size
fl = B * tiles * (L * (2 * T * (4 * H * H + 2 * H * I) + 4 * T * T * H) + 2 * vc.num_patches * (3 * 14 * 14) * H)
res = {"vision_tower_ms": t_vis * 1e3, "vision_tower_tflops": fl / t_vis / 1e12, "tiles": B * tiles}
# warm-up pass of the whole path, then the measured one on a fresh cache
for rep in range(2):
    runner.tree_cache.reset(); runner.token_to_kv_pool_allocator.clear(); runner.req_to_token_pool.clear(); eng.running = []
    first = [make(i, i) for i in range(B)]
    t_cold = timed(lambda: eng.prefill(first))
    again = [make(i, B + i) for i in range(B)]
    runs = runner.vision.encoder_runs
    t_warm = timed(lambda: eng.prefill(again))
    assert runner.vision.encoder_runs == runs and all(q.cached_tokens >= tiles * vc.image_feature_len for q in again)
    eng.finish(list(eng.running))
res.update({"image_text_prefill_ms": t_cold * 1e3, "tokens_per_request": len(first[0].origin_input_ids), "requests": B,
            "prefill_tokens_per_s": B * len(first[0].origin_input_ids) / t_cold,
            "same_image_second_question_ms": t_warm * 1e3, "radix_hit_tokens_per_request": again[0].cached_tokens})
print(json.dumps(res))
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/llava_prefill.json").write_text(json.dumps(res, indent=1))
