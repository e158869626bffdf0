"""SiluAndMul (reference: /root/reference/python/sglang/srt/layers/activation.py:130-150).

`SiluAndMul.forward` is also what plugin.load() registers as the out-of-tree forward of the reference's SiluAndMul: inputs the gfx950
kernel does not take (another dtype than bf16, a width or a layout outside its 16-byte vectors, host tensors) go to the bound
instance's own `forward_native`."""
from __future__ import annotations

import torch
from torch import nn

from .. import kernels

served = dict(hip=0, native=0)


def rows_vectorisable(t: torch.Tensor, width: int) -> bool:
    """A [..., width] bf16 tensor the elementwise kernels can walk as rows of 16-byte vectors without a copy that would detach an
    in-place update: last dimension contiguous, width a multiple of 8 elements, 16-byte aligned base, and either 2-D with a row
    stride of whole vectors or contiguous."""
    if t.dim() == 0 or t.shape[-1] != width or width % 8 != 0 or t.stride(-1) != 1 or t.data_ptr() % 16 != 0:
        return False
    if t.dim() == 1:
        return True
    if t.dim() == 2:
        return t.stride(0) % 8 == 0 or t.shape[0] == 1
    return t.is_contiguous()


class SiluAndMul(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        native = getattr(self, "forward_native", None)
        if native is not None and not (x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 16 == 0
                                       and rows_vectorisable(x, x.shape[-1])):
            served["native"] += 1
            return native(x)
        served["hip"] += 1
        return kernels.silu_and_mul(x)
