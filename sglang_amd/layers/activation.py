"""SiluAndMul (reference: /root/reference/python/sglang/srt/layers/activation.py:130-150)."""
from __future__ import annotations

import torch
from torch import nn

from .. import kernels


class SiluAndMul(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return kernels.silu_and_mul(x)
