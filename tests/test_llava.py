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


# ----------------------------------------------------------------------------------------------- LLaVA-1.6 anyres
# tests/golden/llava_anyres.pt: the reference's OWN pad_input_ids and the packing + substitution of
# LlavaBaseForCausalLM.forward (gen_golden.py gen_llava_anyres: vision tower and language model replaced by recorders).
def _anyres(golden_dir):
    return torch.load(golden_dir / "llava_anyres.pt", weights_only=False)


def test_anyres_grid_and_unpad_shapes_match_the_reference(golden_dir):
    from sglang_amd.harness import llava as L

    d = _anyres(golden_dir)
    S, side = d["image_size"], d["image_size"] // d["patch_size"]
    for c in d["grid_cases"]:
        gw, gh = c["grid"]
        assert L.get_anyres_image_grid_shape(tuple(c["size"]), d["pinpoints"], S) == (gw, gh), c
        assert L.unpad_image_shape(gh * side, gw * side, tuple(c["size"])) == tuple(c["unpad"]), c
        assert ov.anyres_grid(tuple(c["size"]), d["pinpoints"], S) == (gw, gh)


def test_anyres_pad_input_ids_and_feature_packing_match_the_reference(golden_dir):
    """Product (harness/llava.py) and oracle (oracle/vision.py) against the reference's padded prompt, offsets, pad
    lengths and the embeddings it hands to the language model -- including requests whose image starts inside the
    radix-cached prefix (prefix_len 7 / 1500: only the uncached slice of the image is written)."""
    from sglang_amd.harness import llava as L

    d = _anyres(golden_dir)
    S, P, H = d["image_size"], d["patch_size"], d["hidden"]
    side = S // P
    table, nl = d["embed_table"], d["image_newline"]
    for c in d["cases"]:
        size = tuple(c["size"])
        item = L.MultimodalItem(torch.zeros((c["tiles"], 3, 2, 2)), pad_value=c["padded"][c["offsets"][0]], image_size=size)
        padded = L.pad_input_ids(c["prompt"], d["image_token_index"], [item], side * side, d["pinpoints"], S)
        assert padded == c["padded"] and [item.offset] == c["offsets"] and [item.length] == c["pad_len"], size
        assert ov.anyres_len(size, d["pinpoints"], S, side) == c["pad_len"][0]
        packed = L.pack_anyres_features(c["tile_features"], size, d["pinpoints"], S, nl)
        assert torch.equal(packed, ov.pack_anyres(c["tile_features"], size, d["pinpoints"], S, nl)) and packed.shape == (c["pad_len"][0], H)
        # the substitution over the extend range (mm_utils.py:463-503 / llava.py:398-452), through embed_mm_inputs
        vision = type("V", (), {"encode_item": staticmethod(lambda it, packed=packed: packed)})()
        pre = c["prefix_len"]
        ext_ids = torch.tensor(padded[pre:], dtype=torch.int64)
        got = L.embed_mm_inputs(ext_ids, table, [[item]], [pre], [len(padded) - pre], vision)
        assert torch.equal(got, c["input_embeds"]), (size, pre)


@pytest.mark.gpu
def test_llava16_anyres_prefill_at_model_shapes_matches_oracle(device):
    """BASELINE configs[4] at its own shapes: CLIP ViT-L/14-336 (24 layers, hidden 1024; hidden_states[-2]), a 640 x 480
    image = base tile + 2 x 2 anyres grid = 5 tiles -> 2340 image tokens after spatial unpad + newlines, the
    LLaVA-1.6-7B language model's shapes (two of its 32 layers).  The product (tower attention on the gfx950 extend
    kernel, projector, anyres packing, pad-value radix keys, embedding substitution) against the oracle's plain torch
    ops; a second question about the same image prefills only its text (no encoder run, 2349-token radix hit)."""
    import dataclasses

    from oracle.model import OracleLM, weights_from_product_model
    from sglang_amd.harness.engine import Engine, ModelRunner, Req
    from sglang_amd.harness.llava import LLAVA16_GRID_PINPOINTS, ClipVisionConfig, MultimodalItem, pad_input_ids
    from sglang_amd.harness.models import CONFIGS

    cfg = dataclasses.replace(CONFIGS["llava-1.6-7b"], num_hidden_layers=2, name="llava-1.6-7b-2layers")
    vc = ClipVisionConfig(image_grid_pinpoints=LLAVA16_GRID_PINPOINTS)
    runner = ModelRunner(cfg, max_total_tokens=3 * 2400 + 512, max_running_requests=4, max_context_len=2400, device=device,
                         use_graph=False, vision_config=vc)
    eng = Engine(runner)
    g = torch.Generator().manual_seed(11)
    rnd = random.Random(11)
    IMG = cfg.vocab_size + 5
    size = (640, 480)
    tiles = torch.randn((5, 3, vc.image_size, vc.image_size), generator=g)
    head = [rnd.randrange(cfg.vocab_size) for _ in range(9)]

    def prompt(tail):
        items = [MultimodalItem(tiles.clone(), image_size=size)]
        ids = pad_input_ids(head + [IMG] + [rnd.randrange(cfg.vocab_size) for _ in range(tail)], IMG, items, vc.image_feature_len,
                            LLAVA16_GRID_PINPOINTS, vc.image_size)
        return ids, items

    specs = [prompt(6), prompt(4)]
    assert specs[0][1][0].length == 2340 and specs[0][1][0].offset == 9
    new_tokens = 3
    reqs = []
    for i, (ids, items) in enumerate(specs):
        q = Req(i, ids, new_tokens)
        q.mm_items = items
        reqs.append(q)
    eng.logits_by_req = {}
    eng.prefill([reqs[0]])
    assert runner.vision.encoder_runs == 5
    eng.prefill([reqs[1]])
    assert reqs[1].cached_tokens == 9 + 2340 and runner.vision.encoder_runs == 5      # the repeated image: a radix hit, no tower run
    for _ in range(new_tokens - 1):
        eng.decode_step()
    eng.finish(list(eng.running))
    # oracle: vision restatement (plain torch ops, here on the GPU) + anyres packing + the language model on the embeddings
    vw = {k: v.to(device) for k, v in ov.weights_from_product_vision(runner.vision).items()}
    lw = weights_from_product_model(runner.model, device)
    feats = ov.encode_images(tiles.to(device).to(torch.bfloat16).float(), vw, patch=vc.patch_size, heads=vc.num_attention_heads,
                             eps=vc.layer_norm_eps, n_layers=runner.vision.tower.n_layers)
    packed = ov.pack_anyres(feats.view(5, vc.image_feature_len, -1), size, LLAVA16_GRID_PINPOINTS, vc.image_size,
                            runner.vision.image_newline.data.float())
    embeds = [ov.embed_with_images(ids, lw["embed_tokens"].cpu(), [dict(offset=9, features=packed.cpu())]).to(device) for ids, _ in specs]
    oracle = OracleLM(cfg, lw, num_slots=2 * 2400, max_ctx=2400, max_reqs=2, device=device, compute_dtype=torch.float32)
    _, ref = oracle.generate([s[0] for s in specs], new_tokens, return_logits=True, forced=[q.output_ids for q in reqs], prompt_embeds=embeds)
    worst, ref_rms = 0.0, 0.0
    for b, q in enumerate(reqs):
        for k, row in enumerate(eng.logits_by_req[q.rid]):
            r = ref[k][b].float().cpu()
            worst = max(worst, float((row - r).abs().max()))
            ref_rms = max(ref_rms, float(r.pow(2).mean().sqrt()))
    # bf16 tower (24 layers) + projector + two decoder layers against an fp32 tower: a few bf16 ulps of the logits
    assert worst <= 0.1 * max(ref_rms, 1.0), (worst, ref_rms)
