"""Restatement of the LLaVA image path (plain torch ops).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/python/sglang/srt/models/clip.py:51-94,143-300,430-486 (CLIPVisionTransformer: patch
Conv2d(k = s = patch, bias=False) + class token + learned positions, pre_layrnorm, pre-LN encoder layers with
quick_gelu MLPs), models/llava.py:145-168 (encode_images: hidden_states[mm_vision_select_layer], class token dropped,
linear -> GELU -> linear projector) and managers/mm_utils.py:463-503 (clamped embedding + scatter over the pad-value
positions).  The reference's CLIP classes mirror `transformers.CLIPVisionModel` weight for weight, so the
restatement is pinned against that implementation (tests/test_llava.py) -- transformers is a third-party package
present in the image, no reference import is needed.

Weights: a dict with the product tower's names (sglang_amd/harness/llava.py): fused qkv [3H, H] = q | k | v rows.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


def clip_vision_hidden(pixel_values: torch.Tensor, w: Dict[str, torch.Tensor], *, patch: int, heads: int, eps: float,
                       n_layers: int, compute_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """hidden_states[n_layers] of CLIPVisionTransformer: [n, 1 + patches, H]."""
    x = pixel_values.to(compute_dtype)
    n, C, S, _ = x.shape
    H = w["class_embedding"].shape[0]
    conv_w = w["patch_weight"].to(compute_dtype).view(H, C, patch, patch)
    pe = F.conv2d(x, conv_w, stride=patch).flatten(2).transpose(1, 2)                 # [n, patches, H]
    h = torch.cat([w["class_embedding"].to(compute_dtype).expand(n, 1, H), pe], dim=1) + w["position_embedding"].to(compute_dtype)
    h = F.layer_norm(h, (H,), w["pre_ln_w"].to(compute_dtype), w["pre_ln_b"].to(compute_dtype), eps)
    T, D = h.shape[1], H // heads
    for i in range(n_layers):
        g = lambda k: w[f"layers.{i}.{k}"].to(compute_dtype)
        y = F.layer_norm(h, (H,), g("ln1_w"), g("ln1_b"), eps)
        q, k, v = F.linear(y, g("qkv_w"), g("qkv_b")).split(H, dim=-1)
        sh = lambda t: t.reshape(n, T, heads, D).transpose(1, 2)
        a = F.scaled_dot_product_attention(sh(q), sh(k), sh(v), scale=D ** -0.5).transpose(1, 2).reshape(n, T, H)
        h = h + F.linear(a, g("o_w"), g("o_b"))
        y = F.layer_norm(h, (H,), g("ln2_w"), g("ln2_b"), eps)
        y = F.linear(y, g("fc1_w"), g("fc1_b"))
        y = y * torch.sigmoid(1.702 * y)                                                # QuickGELU
        h = h + F.linear(y, g("fc2_w"), g("fc2_b"))
    return h


def encode_images(pixel_values: torch.Tensor, w: Dict[str, torch.Tensor], *, patch: int, heads: int, eps: float, n_layers: int,
                  drop_cls: bool = True, compute_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """llava.py:145-168: [tiles * feature_len, text_hidden]."""
    hs = clip_vision_hidden(pixel_values, w, patch=patch, heads=heads, eps=eps, n_layers=n_layers, compute_dtype=compute_dtype)
    if drop_cls:
        hs = hs[:, 1:]
    y = F.linear(hs, w["proj_w1"].to(compute_dtype), w["proj_b1"].to(compute_dtype))
    y = F.linear(F.gelu(y), w["proj_w2"].to(compute_dtype), w["proj_b2"].to(compute_dtype))
    return y.reshape(-1, y.shape[-1])


def anyres_grid(image_size, pinpoints, tile: int):
    """multimodal/mm_utils.py:114-151,211-248 (select_best_resolution + get_anyres_image_grid_shape): (tiles across, down)."""
    ow, oh = image_size
    best_fit, max_eff, min_waste = None, 0, float("inf")
    for width, height in pinpoints:
        scale = min(width / ow, height / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = width * height - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best_fit = eff, waste, (width, height)
    return best_fit[0] // tile, best_fit[1] // tile


def unpad(t: torch.Tensor, image_size) -> torch.Tensor:
    """mm_utils.py:341-369 unpad_image on a [C, H, W] map."""
    ow, oh = image_size
    ch, cw = t.shape[1:]
    if ow / oh > cw / ch:
        nh = int(oh * (cw / ow))
        p = (ch - nh) // 2
        return t[:, p: ch - p, :]
    nw = int(ow * (ch / oh))
    p = (cw - nw) // 2
    return t[:, :, p: cw - p]


def pack_anyres(feats: torch.Tensor, image_size, pinpoints, tile: int, image_newline: torch.Tensor) -> torch.Tensor:
    """models/llava.py:251-358 ("spatial_unpad", anyres): [1 + gw * gh, side^2, H] -> [len, H]."""
    base, rest = feats[0], feats[1:]
    side = int(round(base.shape[0] ** 0.5))
    gw, gh = anyres_grid(image_size, pinpoints, tile)
    x = rest.view(gh, gw, side, side, -1).permute(4, 0, 2, 1, 3).contiguous().flatten(1, 2).flatten(2, 3)      # [H, gh*side, gw*side]
    x = unpad(x, image_size)
    x = torch.cat((x, image_newline.to(x.dtype)[:, None, None].expand(*x.shape[:-1], 1)), dim=-1)
    return torch.cat((base, x.flatten(1, 2).transpose(0, 1)), dim=0)


def anyres_len(image_size, pinpoints, tile: int, side: int) -> int:
    """models/llava.py:96-127: pad tokens of an anyres image."""
    gw, gh = anyres_grid(image_size, pinpoints, tile)
    ow, oh = image_size
    h, w = gh * side, gw * side
    if ow / oh > w / h:
        nh = int(oh * (w / ow)); p = (h - nh) // 2; nh, nw = h - 2 * p, w
    else:
        nw = int(ow * (h / oh)); p = (w - nw) // 2; nh, nw = h, w - 2 * p
    return side * side + nh * (nw + 1)


def embed_with_images(input_ids: Sequence[int], embed_weight: torch.Tensor, images: List[dict]) -> torch.Tensor:
    """mm_utils.py:463-503 for ONE whole prompt: clamp, embed, overwrite [offset, offset + len) of every image."""
    ids = torch.tensor(list(input_ids), dtype=torch.int64).clamp(0, embed_weight.shape[0] - 1)
    e = F.embedding(ids, embed_weight).clone()
    for im in images:
        e[im["offset"]: im["offset"] + im["features"].shape[0]] = im["features"].to(e.dtype)
    return e


def weights_from_product_vision(vision) -> Dict[str, torch.Tensor]:
    t, p = vision.tower, vision.projector
    w = {"patch_weight": t.patch_weight, "class_embedding": t.class_embedding, "position_embedding": t.position_embedding,
         "pre_ln_w": t.pre_ln_w, "pre_ln_b": t.pre_ln_b, "proj_w1": p.w1, "proj_b1": p.b1, "proj_w2": p.w2, "proj_b2": p.b2}
    for i, m in enumerate(t.layers):
        for k in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "qkv_w", "qkv_b", "o_w", "o_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b"):
            w[f"layers.{i}.{k}"] = getattr(m, k)
    return {k: v.detach().float().cpu() for k, v in w.items()}


def weights_from_hf_clip(model) -> Dict[str, torch.Tensor]:
    """transformers.CLIPVisionModel state -> the names above (q / k / v rows fused)."""
    sd = model.state_dict()
    pre = "vision_model." if any(k.startswith("vision_model.") for k in sd) else ""      # key prefix differs across transformers versions
    H = sd[pre + "embeddings.class_embedding"].shape[0]
    w = {"patch_weight": sd[pre + "embeddings.patch_embedding.weight"].reshape(H, -1),
         "class_embedding": sd[pre + "embeddings.class_embedding"],
         "position_embedding": sd[pre + "embeddings.position_embedding.weight"],
         "pre_ln_w": sd[pre + "pre_layrnorm.weight"], "pre_ln_b": sd[pre + "pre_layrnorm.bias"]}
    i = 0
    while f"{pre}encoder.layers.{i}.layer_norm1.weight" in sd:
        L = f"{pre}encoder.layers.{i}."
        w[f"layers.{i}.ln1_w"], w[f"layers.{i}.ln1_b"] = sd[L + "layer_norm1.weight"], sd[L + "layer_norm1.bias"]
        w[f"layers.{i}.ln2_w"], w[f"layers.{i}.ln2_b"] = sd[L + "layer_norm2.weight"], sd[L + "layer_norm2.bias"]
        w[f"layers.{i}.qkv_w"] = torch.cat([sd[L + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        w[f"layers.{i}.qkv_b"] = torch.cat([sd[L + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        w[f"layers.{i}.o_w"], w[f"layers.{i}.o_b"] = sd[L + "self_attn.out_proj.weight"], sd[L + "self_attn.out_proj.bias"]
        w[f"layers.{i}.fc1_w"], w[f"layers.{i}.fc1_b"] = sd[L + "mlp.fc1.weight"], sd[L + "mlp.fc1.bias"]
        w[f"layers.{i}.fc2_w"], w[f"layers.{i}.fc2_b"] = sd[L + "mlp.fc2.weight"], sd[L + "mlp.fc2.bias"]
        i += 1
    return {k: v.detach().float() for k, v in w.items()}
