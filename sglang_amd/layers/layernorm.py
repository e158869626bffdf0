"""RMSNorm with SGLang's module signature, running the gfx950 kernels
(reference: /root/reference/python/sglang/srt/layers/layernorm.py:423-826).

`RMSNorm.forward` is also what plugin.load() registers as the out-of-tree forward of the reference's RMSNorm
(BaseFusedOp.register_oot_forward): bound to a reference instance it reads the same attributes (`weight`,
`variance_epsilon`, and the optional `variance_size_override` / `cast_x_before_out_mul` / `fp32_residual` /
`override_orig_dtype` / `x_pad_to_multiple` switches) and hands every configuration outside the bf16 hot path to that instance's own `forward_native`."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import nn

from .. import kernels
from .activation import rows_vectorisable

# calls served by the gfx950 kernels / handed to the bound reference instance's own forward_native (tests and the
# reference-stack reports read it: a silent hand-over is a slower path, not an error)
served = dict(hip=0, native=0)


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype: torch.dtype = torch.bfloat16, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype, device=device), requires_grad=False)
        self.variance_epsilon = eps
        self.hidden_size = hidden_size

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None,
                post_residual_addition: Optional[torch.Tensor] = None, quant_linear: Optional[nn.Module] = None
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        """layernorm.py:474-480 signature.  `x` (and `residual`) are updated IN PLACE when a residual is given, like
        sgl_kernel.fused_add_rmsnorm (layernorm.py:739-751); `post_residual_addition` is folded into the residual
        first (layernorm.py:563-564: hidden_states + (residual + post_residual_addition))."""
        # `quant_linear` is only a HINT (the projection that consumes the output: llama.py:348-366 passes it on every call): the
        # reference fuses an fp8 activation quantisation into the norm when that projection carries a static per-tensor
        # input scale (layernorm.py:372-392, 508-521) and ignores it otherwise.  Round 4 treated ANY quant_linear as outside
        # the path -- so under the reference's own LlamaDecoderLayer every prefill norm ran as eight torch launches of
        # forward_native (found in round 5's kernel trace of the reference scheduler, profiles/r05_sched_kernel_stats_*).
        fp8_fusion = quant_linear is not None and getattr(quant_linear, "input_scale", None) is not None
        outside = (fp8_fusion or x.dtype != torch.bfloat16 or self.weight.dtype != torch.bfloat16
                   or getattr(self, "variance_size_override", None) is not None
                   or getattr(self, "cast_x_before_out_mul", False) or getattr(self, "fp32_residual", False)
                   or getattr(self, "override_orig_dtype", None) is not None or (getattr(self, "x_pad_to_multiple", 0) or 0) > 0
                   or not x.is_cuda or x.shape[-1] > 65536 or not rows_vectorisable(x, x.shape[-1])
                   or self.weight.shape[-1] != x.shape[-1] or self.weight.data_ptr() % 16 != 0
                   or (residual is not None and (residual.dtype != x.dtype or residual.shape != x.shape
                                                 or not rows_vectorisable(residual, x.shape[-1]))))
        if outside:
            native = getattr(self, "forward_native", None)
            if native is None:
                raise NotImplementedError("RMSNorm: this configuration (quant_linear / non-bf16 / variance override) is "
                                          "outside the gfx950 path")
            served["native"] += 1
            return native(x, residual, post_residual_addition, quant_linear)
        served["hip"] += 1
        if x.numel() == 0:                                   # layernorm.py:481-486
            if residual is not None:
                if post_residual_addition is not None:
                    residual = residual + post_residual_addition
                return x, residual
            return x
        if residual is not None:
            if post_residual_addition is not None:
                residual = residual + post_residual_addition
            kernels.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        return kernels.rmsnorm(x, self.weight.data, self.variance_epsilon)
