"""BASELINE configs[4] / SURVEY section 8(f2): LLaVA image + text prefill.
CPU: the oracle's CLIP restatement pinned against transformers' CLIPVisionModel (the implementation the reference's
clip.py mirrors), pad values and pad_input_ids, and what they do to the radix tree.  GPU: the product path (vision
tower with its attention on the gfx950 extend kernel, projector, embedding substitution, radix hit on a repeated
image) against the oracle."""
import random

import pytest
import torch

from oracle import vision as ov


def test_clip_restatement_matches_transformers_clip_vision_model():
    transformers = pytest.importorskip("transformers")
    cfg = transformers.CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                                        image_size=28, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    torch.manual_seed(0)
    hf = transformers.CLIPVisionModel(cfg).eval()
    x = torch.randn(2, 3, 28, 28)
    with torch.no_grad():
        want = hf(pixel_values=x, output_hidden_states=True).hidden_states
    w = ov.weights_from_hf_clip(hf)
    for k in (0, 2, 3):                                           # hidden_states[-2] is what LLaVA takes
        got = ov.clip_vision_hidden(x, w, patch=14, heads=2, eps=1e-5, n_layers=k)
        torch.testing.assert_close(got, want[k], atol=2e-5, rtol=2e-5)


def test_pad_values_and_padded_prompts_drive_the_radix_tree():
    from sglang_amd.harness.engine import Req
    from sglang_amd.harness.llava import MM_PAD_SHIFT_VALUE, MultimodalItem, compute_pad_value, pad_input_ids
    from sglang_amd.mem_cache.allocator import TokenToKVPoolAllocator
    from sglang_amd.mem_cache.memory_pool import ReqToTokenPool
    from sglang_amd.mem_cache.radix_cache import InsertParams, MatchPrefixParams, RadixCache, RadixKey

    assert compute_pad_value(5) == MM_PAD_SHIFT_VALUE + 5 and compute_pad_value((1 << 30) + 7) == MM_PAD_SHIFT_VALUE + 7
    g = torch.Generator().manual_seed(1)
    img_a, img_b = torch.randn((1, 3, 28, 28), generator=g), torch.randn((1, 3, 28, 28), generator=g)
    IMG = 32000                                                   # image_token_index
    text = [1, 5, 6, IMG, 9, 10]
    items_a, items_a2, items_b = [MultimodalItem(img_a)], [MultimodalItem(img_a.clone())], [MultimodalItem(img_b)]
    pa = pad_input_ids(text, IMG, items_a, 4)
    pa2 = pad_input_ids(text[:5] + [77], IMG, items_a2, 4)        # the same image, another question
    pb = pad_input_ids(text, IMG, items_b, 4)
    assert items_a[0].pad_value == items_a2[0].pad_value != items_b[0].pad_value and items_a[0].pad_value >= MM_PAD_SHIFT_VALUE
    assert pa == [1, 5, 6] + [items_a[0].pad_value] * 4 + [9, 10] and (items_a[0].offset, items_a[0].length) == (3, 4)
    dev = torch.device("cpu")
    pool = ReqToTokenPool(4, 32, dev)
    alloc = TokenToKVPoolAllocator(64, torch.bfloat16, dev, None)
    tree = RadixCache(pool, alloc, 1)
    tree.insert(InsertParams(key=RadixKey(Req(0, pa, 1).origin_array), value=torch.arange(1, len(pa) + 1)))
    hit = lambda ids: int(tree.match_prefix(MatchPrefixParams(key=RadixKey(Req(1, ids, 1).origin_array))).device_indices.numel())
    assert hit(pa2) == 3 + 4 + 1            # text before the image, the whole image, one shared text token
    assert hit(pb) == 3                     # another image: the prefix ends where the pixels differ


@pytest.mark.gpu
def test_llava_image_text_prefill_matches_oracle(device):
    from oracle.model import OracleLM, weights_from_product_model
    from sglang_amd.harness.engine import Engine, ModelRunner, Req
    from sglang_amd.harness.llava import TINY_CLIP, MultimodalItem, pad_input_ids
    from sglang_amd.harness.models import CONFIGS

    cfg = CONFIGS["tiny-llama"]
    vc = TINY_CLIP
    runner = ModelRunner(cfg, max_total_tokens=4096, max_running_requests=8, max_context_len=256, device=device,
                         init_device="cpu", use_graph=True, vision_config=vc)
    eng = Engine(runner)
    g = torch.Generator().manual_seed(3)
    rnd = random.Random(3)
    IMG = cfg.vocab_size + 5                                     # any id that is not a text token
    img1 = torch.randn((2, 3, vc.image_size, vc.image_size), generator=g)      # two tiles (anyres-style: tiles x 16 tokens)
    img2 = torch.randn((1, 3, vc.image_size, vc.image_size), generator=g)
    flen = vc.image_feature_len
    head = [rnd.randrange(cfg.vocab_size) for _ in range(7)]
    def prompt(img, tail):
        items = [MultimodalItem(img)]
        return pad_input_ids(head + [IMG] + [rnd.randrange(cfg.vocab_size) for _ in range(tail)], IMG, items, flen), items
    specs = [prompt(img1, 9), prompt(img1, 5), prompt(img2, 6), (head + [rnd.randrange(cfg.vocab_size) for _ in range(11)], None)]
    new_tokens = 4
    reqs = []
    for i, (ids, items) in enumerate(specs):
        q = Req(i, ids, new_tokens)
        q.mm_items = items
        reqs.append(q)
    eng.logits_by_req = {}
    eng.prefill([reqs[0]])
    runs_after_first = runner.vision.encoder_runs
    assert runs_after_first == 2
    eng.prefill(reqs[1:])
    # request 1 repeats image 1: head + the whole image come from the radix cache, its tiles are NOT encoded again;
    # request 2 has a different image behind the same head: the hit stops at the first pixel-derived id
    assert reqs[1].cached_tokens == 7 + 2 * flen and reqs[2].cached_tokens == 7 and reqs[3].cached_tokens == 7
    assert runner.vision.encoder_runs == runs_after_first + 1
    for _ in range(new_tokens - 1):
        eng.decode_step()
    eng.finish(list(eng.running))
    # oracle: the vision path restated on the CPU, then the language model on the substituted embeddings
    vw = ov.weights_from_product_vision(runner.vision)
    lw = weights_from_product_model(runner.model)
    embeds = []
    for ids, items in specs:
        if items is None:
            embeds.append(None)
            continue
        feats = ov.encode_images(items[0].pixel_values.to(torch.bfloat16).float(), vw, patch=vc.patch_size, heads=vc.num_attention_heads,
                                 eps=vc.layer_norm_eps, n_layers=runner.vision.tower.n_layers)
        embeds.append(ov.embed_with_images(ids, lw["embed_tokens"], [dict(offset=items[0].offset, features=feats)]))
    oracle = OracleLM(cfg, lw, compute_dtype=torch.float32)
    _, ref = oracle.generate([s[0] for s in specs], new_tokens, return_logits=True, forced=[q.output_ids for q in reqs],
                             prompt_embeds=embeds)
    for b, q in enumerate(reqs):
        for k, row in enumerate(eng.logits_by_req[q.rid]):
            torch.testing.assert_close(row, ref[k][b], atol=4e-2, rtol=4e-2, msg=f"request {q.rid} token {k}")
