"""LLaVA image + text prefill on the hot path: CLIP ViT vision tower, projector, pad-value radix keys and the
embedding substitution (BASELINE configs[4]: LLaVA-1.6-7B image+text prefill).

Reference (/root/reference/python/sglang/srt):
  models/clip.py:51-94 (patch + class + position embeddings), :143-300 (pre-LN encoder layer: LayerNorm -> MHA ->
  residual, LayerNorm -> fc1 -> quick_gelu -> fc2 -> residual), :430-486 (pre_layrnorm, encoder);
  models/llava.py:79-143 pad_input_ids (the image token becomes `image_feature_len` copies of the image's pad value;
  LLaVA-1.6 "anyres": base tile + the unpadded high-resolution grid with one image_newline column per row),
  :145-168 encode_images (hidden state of `mm_vision_select_layer` = -2, CLS dropped, projector linear-GELU-linear),
  :251-358 anyres feature packing (grid view, spatial unpad, image_newline, base tile first);
  multimodal/mm_utils.py:114-151 select_best_resolution, :211-248 get_anyres_image_grid_shape, :341-393 unpad_image(_shape);
  managers/schedule_batch.py:147-148,220-222 pad value = 1_000_000 + hash % 2^30;
  managers/mm_utils.py:463-503 embed_mm_inputs (clamp ids into the vocabulary, embed, scatter the image features over
  the pad-value positions of the EXTEND range -- image tokens that sit in the radix-cached prefix need no encoder run).

What runs where on MI355X: the tower's and the projector's matmuls are plain library GEMMs (hipBLASLt through torch);
its bidirectional attention runs on the gfx950 extend-attention kernel (non-causal, the image's K/V rows in a scratch
pool); the multimodal part of the path that is NEW relative to text is integer work on the host: pad values make two
different images differ in their radix keys and make the same image a prefix hit, so a second question about an
image prefills only its text.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import kernels
from .models import BF, synth_weight

MM_PAD_SHIFT_VALUE = 1_000_000          # schedule_batch.py:147


def compute_pad_value(hash_: int) -> int:
    """schedule_batch.py:220-222."""
    return MM_PAD_SHIFT_VALUE + (hash_ % (1 << 30))


def hash_pixels(pixel_values: torch.Tensor) -> int:
    """A content hash of the image tensor (the reference hashes the raw feature bytes: mm_utils.py hash_feature)."""
    return zlib.crc32(pixel_values.detach().to(torch.float32).cpu().numpy().tobytes()) | (pixel_values.numel() << 32)


@dataclass
class ClipVisionConfig:
    """CLIP ViT-L/14-336 (the LLaVA-1.5 / 1.6 tower) by default."""

    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 336
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-5
    select_layer: int = -2            # mm_vision_select_layer
    select_feature: str = "patch"     # drop the class token
    image_grid_pinpoints: Optional[List[List[int]]] = None     # LLaVA-1.6: candidate canvases of the anyres grid

    @property
    def num_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2

    @property
    def image_feature_len(self) -> int:
        return self.num_patches if self.select_feature == "patch" else self.num_patches + 1


TINY_CLIP = ClipVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                             image_size=56, patch_size=14)


LLAVA16_GRID_PINPOINTS = [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]]    # llava-v1.6-*-7b config.json


def select_best_resolution(original_size: Tuple[int, int], resolutions: Sequence[Sequence[int]]) -> Tuple[int, int]:
    """mm_utils.py:114-151: the candidate (width, height) that keeps most of the image's pixels after an
    aspect-preserving downscale; ties go to the one that wastes the least canvas."""
    ow, oh = original_size
    best, best_eff, best_waste = None, 0, float("inf")
    for w, h in resolutions:
        scale = min(w / ow, h / oh)
        eff = min(int(ow * scale) * int(oh * scale), ow * oh)
        waste = w * h - eff
        if eff > best_eff or (eff == best_eff and waste < best_waste):
            best, best_eff, best_waste = (w, h), eff, waste
    return best


def get_anyres_image_grid_shape(image_size: Tuple[int, int], grid_pinpoints: Sequence[Sequence[int]], tile_size: int) -> Tuple[int, int]:
    """mm_utils.py:211-248 for an explicit pinpoint list: (tiles across, tiles down) of the high-resolution grid."""
    w, h = select_best_resolution(image_size, grid_pinpoints)
    return w // tile_size, h // tile_size


def _unpad_bounds(cur_h: int, cur_w: int, original_size: Tuple[int, int]) -> Tuple[int, int, int, int]:
    """(row0, row1, col0, col1) of the part of a [cur_h, cur_w] feature map that the resized image covers
    (mm_utils.py:341-393: the image was scaled to fit and centred; the padding bands carry no image)."""
    ow, oh = original_size
    if ow / oh > cur_w / cur_h:                       # wider than the canvas: bands above and below
        new_h = int(oh * (cur_w / ow))
        pad = (cur_h - new_h) // 2
        return pad, cur_h - pad, 0, cur_w
    new_w = int(ow * (cur_h / oh))
    pad = (cur_w - new_w) // 2
    return 0, cur_h, pad, cur_w - pad


def unpad_image_shape(cur_h: int, cur_w: int, original_size: Tuple[int, int]) -> Tuple[int, int]:
    r0, r1, c0, c1 = _unpad_bounds(cur_h, cur_w, original_size)
    return r1 - r0, c1 - c0


def anyres_feature_len(image_size: Tuple[int, int], grid_pinpoints, tile_size: int, side: int) -> int:
    """llava.py:96-127: base tile (side^2 features) + the unpadded grid, every row closed by one image_newline."""
    gw, gh = get_anyres_image_grid_shape(image_size, grid_pinpoints, tile_size)
    nh, nw = unpad_image_shape(gh * side, gw * side, image_size)
    return side * side + nh * (nw + 1)


def pack_anyres_features(feats: torch.Tensor, image_size: Tuple[int, int], grid_pinpoints, tile_size: int,
                         image_newline: torch.Tensor) -> torch.Tensor:
    """llava.py:251-358 ("spatial_unpad"): feats [1 + gw * gh, side^2, H] -> [feature_len, H].  The grid tiles are laid
    out as one [gh * side, gw * side] map, cropped to the part the image covers, each row gets the image_newline
    embedding appended, rows are concatenated; the base (whole-image) tile goes first."""
    tiles, n, hid = feats.shape
    side = int(round(n ** 0.5))
    gw, gh = get_anyres_image_grid_shape(image_size, grid_pinpoints, tile_size)
    if tiles != 1 + gw * gh or side * side != n:
        raise ValueError(f"anyres: {tiles} tiles of {n} features for a {gw} x {gh} grid of {side} x {side} tiles")
    grid = feats[1:].view(gh, gw, side, side, hid).permute(0, 2, 1, 3, 4).reshape(gh * side, gw * side, hid)
    r0, r1, c0, c1 = _unpad_bounds(gh * side, gw * side, image_size)
    grid = grid[r0:r1, c0:c1]
    nl = image_newline.to(grid.dtype).view(1, 1, hid).expand(grid.shape[0], 1, hid)
    return torch.cat([feats[0], torch.cat([grid, nl], dim=1).reshape(-1, hid)], dim=0)


@dataclass
class MultimodalItem:
    """One image of a request (schedule_batch.py MultimodalDataItem: feature, pad_value, offsets, image_sizes)."""

    pixel_values: torch.Tensor            # [tiles, 3, S, S]
    pad_value: int = 0
    offset: int = -1                      # first pad position in the padded prompt
    length: int = 0                       # number of pad tokens
    image_size: Optional[Tuple[int, int]] = None   # (width, height) of the original image: LLaVA-1.6 anyres packing
                                                   # (tiles = base + grid); None: fixed tiles x image_feature_len

    def __post_init__(self):
        if not self.pad_value:
            self.pad_value = compute_pad_value(hash_pixels(self.pixel_values))


def pad_input_ids(input_ids: Sequence[int], image_token_index: int, items: List[MultimodalItem], feature_len: int,
                  grid_pinpoints=None, tile_size: Optional[int] = None) -> List[int]:
    """llava.py:79-143: every occurrence of the image token becomes the item's pad value repeated once per image
    feature -- `tiles * feature_len` for fixed tiles, base + unpadded grid + newlines for an anyres item (which needs the
    model's grid pinpoints and tile size); offsets / lengths are recorded on the items."""
    ids = list(input_ids)
    for it in items:
        try:
            off = ids.index(image_token_index)
        except ValueError:
            off = 0
        if it.image_size is not None:
            if grid_pinpoints is None or tile_size is None:
                raise ValueError("pad_input_ids: anyres items need grid_pinpoints and tile_size")
            n = anyres_feature_len(it.image_size, grid_pinpoints, tile_size, int(round(feature_len ** 0.5)))
        else:
            n = int(it.pixel_values.shape[0]) * feature_len
        ids = ids[:off] + [it.pad_value] * n + ids[off + 1:]
        it.offset, it.length = off, n
    return ids


class ClipVisionTower(nn.Module):
    """clip.py CLIPVisionTransformer up to the selected layer, synthetic weights."""

    def __init__(self, cfg: ClipVisionConfig, device, init_device=None, prefix: str = "vision_tower"):
        super().__init__()
        self.cfg = cfg
        self.device = device
        ini = init_device if init_device is not None else device
        H, I, P = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size
        self.n_layers = cfg.num_hidden_layers + 1 + cfg.select_layer if cfg.select_layer < 0 else cfg.select_layer
        # hidden_states[k] = output of layer k (hidden_states[0] = embeddings after pre_layrnorm): -2 -> run L - 1 layers

        def W(name, shape, std=0.02):
            return nn.Parameter(synth_weight(f"{prefix}.{name}", shape, ini, std).to(device), requires_grad=False)

        self.patch_weight = W("patch_embedding", (H, cfg.num_channels * P * P))           # Conv2d(k = s = P, bias=False) as a GEMM
        self.class_embedding = W("class_embedding", (H,))
        self.position_embedding = W("position_embedding", (cfg.num_patches + 1, H))
        self.pre_ln_w, self.pre_ln_b = W("pre_layrnorm.weight", (H,), 0.0), W("pre_layrnorm.bias", (H,), 0.0)
        self.pre_ln_w.data.fill_(1.0)
        self.layers = nn.ModuleList()
        for i in range(self.n_layers):
            m = nn.Module()
            p = f"layers.{i}"
            m.ln1_w, m.ln1_b = W(f"{p}.ln1.w", (H,), 0.0), W(f"{p}.ln1.b", (H,), 0.0)
            m.ln2_w, m.ln2_b = W(f"{p}.ln2.w", (H,), 0.0), W(f"{p}.ln2.b", (H,), 0.0)
            m.ln1_w.data.fill_(1.0); m.ln2_w.data.fill_(1.0)
            m.qkv_w, m.qkv_b = W(f"{p}.qkv.w", (3 * H, H)), W(f"{p}.qkv.b", (3 * H,))
            m.o_w, m.o_b = W(f"{p}.o.w", (H, H)), W(f"{p}.o.b", (H,))
            m.fc1_w, m.fc1_b = W(f"{p}.fc1.w", (I, H)), W(f"{p}.fc1.b", (I,))
            m.fc2_w, m.fc2_b = W(f"{p}.fc2.w", (H, I)), W(f"{p}.fc2.b", (H,))
            self.layers.append(m)
        self._scratch: Dict[Tuple[int, int], Tuple[torch.Tensor, ...]] = {}

    def _attention(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_img: int, T: int) -> torch.Tensor:
        """Bidirectional attention of `n_img` images x T tokens on the gfx950 extend kernel: the images' K/V rows go to
        a scratch pool (slot 1 + t), every image is one request with no prefix and a non-causal mask."""
        Hh, D = self.cfg.num_attention_heads, self.cfg.hidden_size // self.cfg.num_attention_heads
        key = (n_img, T)
        if key not in self._scratch:
            dev = q.device
            r2t = torch.zeros((n_img + 1, T), dtype=torch.int32, device=dev)
            r2t[1:] = (torch.arange(n_img * T, device=dev, dtype=torch.int32) + 1).view(n_img, T)
            self._scratch[key] = (torch.zeros((n_img * T + 1, Hh, D), dtype=BF, device=dev),
                                  torch.zeros((n_img * T + 1, Hh, D), dtype=BF, device=dev), r2t,
                                  torch.arange(1, n_img + 1, device=dev), torch.full((n_img,), T, dtype=torch.int32, device=dev),
                                  torch.zeros(n_img, dtype=torch.int32, device=dev),
                                  (torch.arange(n_img + 1, device=dev) * T).to(torch.int32),
                                  torch.arange(1, n_img * T + 1, device=dev, dtype=torch.int64))
        kc, vc, r2t, pool, seq, pre, qo, loc = self._scratch[key]
        kernels.store_kv_cache(k.contiguous(), v.contiguous(), kc, vc, loc)
        out = torch.empty((n_img * T, Hh, D), dtype=BF, device=q.device)
        kernels.extend_attention(q.contiguous().view(n_img * T, Hh, D), out, kc, vc, r2t, pool, seq, pre, qo, T, D ** -0.5, causal=False)
        return out.view(n_img * T, Hh * D)

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[n, 3, S, S] -> [n, 1 + patches, hidden]: hidden_states[select_layer]."""
        cfg = self.cfg
        n, P, H = pixel_values.shape[0], cfg.patch_size, cfg.hidden_size
        g = cfg.image_size // P
        x = pixel_values.to(self.device, BF)
        patches = x.view(n, cfg.num_channels, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(n * g * g, cfg.num_channels * P * P)
        pe = F.linear(patches, self.patch_weight).view(n, g * g, H)
        h = torch.cat([self.class_embedding.expand(n, 1, H), pe], dim=1) + self.position_embedding
        T = h.shape[1]
        h = F.layer_norm(h, (H,), self.pre_ln_w, self.pre_ln_b, cfg.layer_norm_eps).view(n * T, H)
        for m in self.layers:
            y = F.layer_norm(h, (H,), m.ln1_w, m.ln1_b, cfg.layer_norm_eps)
            qkv = F.linear(y, m.qkv_w, m.qkv_b)
            q, k, v = qkv.split(H, dim=-1)
            if h.is_cuda:
                a = self._attention(q, k, v, n, T)
            else:                                           # host construction / CPU tests of the wiring only
                a = _sdpa(q, k, v, n, T, cfg.num_attention_heads)
            h = h + F.linear(a, m.o_w, m.o_b)
            y = F.layer_norm(h, (H,), m.ln2_w, m.ln2_b, cfg.layer_norm_eps)
            y = F.linear(y, m.fc1_w, m.fc1_b)
            y = y * torch.sigmoid(1.702 * y)                # quick_gelu (clip.py:161)
            h = h + F.linear(y, m.fc2_w, m.fc2_b)
        return h.view(n, T, H)


def _sdpa(q, k, v, n, T, heads):
    D = q.shape[-1] // heads
    def sh(t):
        return t.reshape(n, T, heads, D).transpose(1, 2)
    return F.scaled_dot_product_attention(sh(q), sh(k), sh(v)).transpose(1, 2).reshape(n * T, heads * D)


class LlavaProjector(nn.Module):
    """llava.py mm_projector "mlp2x_gelu": linear_1 -> GELU -> linear_2."""

    def __init__(self, vision_hidden: int, text_hidden: int, device, init_device=None):
        super().__init__()
        ini = init_device if init_device is not None else device
        self.w1 = nn.Parameter(synth_weight("mm_projector.0.weight", (text_hidden, vision_hidden), ini).to(device), requires_grad=False)
        self.b1 = nn.Parameter(synth_weight("mm_projector.0.bias", (text_hidden,), ini).to(device), requires_grad=False)
        self.w2 = nn.Parameter(synth_weight("mm_projector.2.weight", (text_hidden, text_hidden), ini).to(device), requires_grad=False)
        self.b2 = nn.Parameter(synth_weight("mm_projector.2.bias", (text_hidden,), ini).to(device), requires_grad=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.linear(F.gelu(F.linear(x, self.w1, self.b1)), self.w2, self.b2)


class LlavaVision(nn.Module):
    """Vision tower + projector: encode_images (llava.py:145-168)."""

    def __init__(self, vcfg: ClipVisionConfig, text_hidden: int, device, init_device=None):
        super().__init__()
        self.vcfg = vcfg
        self.tower = ClipVisionTower(vcfg, device, init_device)
        self.projector = LlavaProjector(vcfg.hidden_size, text_hidden, device, init_device)
        # language_model.model.image_newline (llava.py:338-345): closes every row of the unpadded anyres grid
        self.image_newline = nn.Parameter(synth_weight("model.image_newline", (text_hidden,), init_device if init_device is not None else device).to(device),
                                          requires_grad=False)
        self.encoder_runs = 0            # tests: how many tiles went through the tower

    def encode_images(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[tiles, 3, S, S] -> [tiles * image_feature_len, text_hidden]."""
        hs = self.tower(pixel_values)
        self.encoder_runs += int(pixel_values.shape[0])
        if self.vcfg.select_feature == "patch":
            hs = hs[:, 1:]
        return self.projector(hs).reshape(-1, self.projector.w2.shape[0])

    def encode_item(self, item: "MultimodalItem") -> torch.Tensor:
        """All features of one image in prompt order: [item.length, text_hidden]."""
        feats = self.encode_images(item.pixel_values)
        if item.image_size is None:
            return feats
        tiles = int(item.pixel_values.shape[0])
        pins = self.vcfg.image_grid_pinpoints or LLAVA16_GRID_PINPOINTS
        return pack_anyres_features(feats.view(tiles, -1, feats.shape[-1]), item.image_size, pins, self.vcfg.image_size, self.image_newline)


def embed_mm_inputs(input_ids: torch.Tensor, embed_weight: torch.Tensor, reqs_items: Sequence[Optional[List[MultimodalItem]]],
                    extend_prefix_lens: Sequence[int], extend_seq_lens: Sequence[int], vision: LlavaVision) -> torch.Tensor:
    """mm_utils.py:463-503 for an extend batch: ids are clamped into the vocabulary and embedded, then every image's
    features are written over ITS pad-value positions inside the request's extend range [prefix, prefix + extend) --
    only the slice of the image that is not already in the radix-cached prefix is needed, and an image that lies
    entirely in the prefix costs no encoder run at all."""
    vocab = embed_weight.shape[0]
    embeds = F.embedding(input_ids.clamp(0, vocab - 1), embed_weight)
    cache: Dict[int, torch.Tensor] = {}                       # the same image in several requests is encoded once
    start = 0
    for items, pre, ext in zip(reqs_items, extend_prefix_lens, extend_seq_lens):
        for it in items or []:
            lo, hi = max(it.offset, pre), min(it.offset + it.length, pre + ext)
            if lo < hi:
                feats = cache.get(it.pad_value)
                if feats is None:
                    feats = cache[it.pad_value] = vision.encode_item(it)
                embeds[start + lo - pre: start + hi - pre] = feats[lo - it.offset: hi - it.offset].to(embeds.dtype)
        start += ext
    return embeds
