"""RMSNorm with SGLang's module signature, running the gfx950 kernels
(reference: /root/reference/python/sglang/srt/layers/layernorm.py:423-826)."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import nn

from .. import kernels


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype: torch.dtype = torch.bfloat16, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype, device=device), requires_grad=False)
        self.variance_epsilon = eps
        self.hidden_size = hidden_size

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        """`x` (and `residual`) are updated IN PLACE when a residual is given, like
        sgl_kernel.fused_add_rmsnorm (layernorm.py:739-751)."""
        if residual is not None:
            kernels.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        return kernels.rmsnorm(x, self.weight.data, self.variance_epsilon)
