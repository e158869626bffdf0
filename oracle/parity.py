"""Teacher-forced logit parity of a whole generation job against the oracle.  TEST INFRASTRUCTURE ONLY
(imported by tests/ and by bench.py's `parity` leg, after its timed region -- never by the product).

The product run records every step's next-token logits (device tensors, model dtype); the oracle
(oracle/model.py, plain torch ops = the reference's torch-native path) is then driven with the product's own
tokens as inputs (`forced`), so step k of both runs sees the same token history and the logits can be
compared position by position even after an arg-max flipped on a near-tie.

Two oracle flavours, as SURVEY.md section 8(c) prescribes:
  fp32acc : attention evaluated in fp32 on the bf16-rounded inputs (compute_dtype=float32)
  literal : the literal bf16 SDPA of torch_native_backend.py
and, for context, the disagreement between those two references themselves.

north_star bar: "bf16 logits within 1e-3".  Logits of these synthetic-weight models reach |x| ~ 6 where one
bf16 ulp is 3.1e-2, so two correct bf16 pipelines differ by whole ulps wherever their pre-rounding values fall on
different sides of a rounding boundary; the report therefore carries the literal 1e-3 fraction AND the
ulp-normalised figures (fraction bit-identical, within 1 / 2 bf16 ulp of the reference value), rms / max,
and arg-max agreement.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .model import OracleLM, weights_from_product_model


def bf16_ulp(x: torch.Tensor) -> torch.Tensor:
    """Size of one bf16 ulp at |x| (8 significand bits): 2^(floor(log2|x|) - 7); subnormal floor 2^-133."""
    e = torch.floor(torch.log2(x.abs().float().clamp_min(2.0 ** -126)))
    return torch.exp2(e - 7.0)


class LogitStats:
    """Streaming comparison of got vs ref logits ([B, V] per step)."""

    def __init__(self, margin: float = 0.0):
        self.n = 0
        self.sum_sq = 0.0
        self.sum_ref_sq = 0.0
        self.max_abs = 0.0
        self.within_1e3 = 0
        self.identical = 0
        self.within_1ulp = 0
        self.within_2ulp = 0
        self.rows = 0
        self.argmax_eq = 0
        self.clear_rows = 0
        self.clear_argmax_eq = 0
        self.margin = margin
        self.worst_step = -1
        self.per_step_rms: List[float] = []

    def update(self, step: int, got: torch.Tensor, ref: torch.Tensor) -> None:
        got, ref = got.float(), ref.float()
        d = (got - ref).abs()
        self.n += d.numel()
        ssq = float(d.pow(2).sum())
        self.sum_sq += ssq
        self.sum_ref_sq += float(ref.pow(2).sum())
        self.per_step_rms.append((ssq / d.numel()) ** 0.5)
        mx = float(d.max())
        if mx > self.max_abs:
            self.max_abs, self.worst_step = mx, step
        self.within_1e3 += int((d <= 1e-3).sum())
        self.identical += int((d == 0).sum())
        ulp = bf16_ulp(ref)
        self.within_1ulp += int((d <= ulp).sum())
        self.within_2ulp += int((d <= 2 * ulp).sum())
        ga, ra = got.argmax(-1), ref.argmax(-1)
        self.rows += got.shape[0]
        self.argmax_eq += int((ga == ra).sum())
        top2 = ref.topk(2, dim=-1).values
        # a "clear" row: the reference's winner leads by more than `margin` bf16 ulps of its own value
        clear = (top2[:, 0] - top2[:, 1]) > self.margin * bf16_ulp(top2[:, 0])
        self.clear_rows += int(clear.sum())
        self.clear_argmax_eq += int((ga[clear] == ra[clear]).sum())

    def summary(self) -> Dict[str, float]:
        n = max(self.n, 1)
        return {
            "logits_compared": self.n,
            "max_abs": self.max_abs,
            "rms": (self.sum_sq / n) ** 0.5,
            "ref_rms": (self.sum_ref_sq / n) ** 0.5,
            "frac_within_1e-3": self.within_1e3 / n,
            "frac_bit_identical": self.identical / n,
            "frac_within_1_bf16_ulp": self.within_1ulp / n,
            "frac_within_2_bf16_ulp": self.within_2ulp / n,
            "argmax_agreement": self.argmax_eq / max(self.rows, 1),
            "argmax_agreement_clear_margin": self.clear_argmax_eq / max(self.clear_rows, 1),
            "clear_margin_rows": self.clear_rows, "rows": self.rows,
            "worst_step": self.worst_step,
            "rms_first_step": self.per_step_rms[0] if self.per_step_rms else None,
            "rms_last_step": self.per_step_rms[-1] if self.per_step_rms else None,
        }


def teacher_forced_parity(cfg, model, prompts: Sequence[Sequence[int]], outs: Sequence[Sequence[int]],
                          step_logits: Sequence[torch.Tensor], *, device=None, flavours=("fp32acc", "literal"),
                          max_ctx: Optional[int] = None, clear_margin_ulps: float = 16.0, batched_decode: bool = True,
                          forced_topk_ids: Optional[Dict[int, torch.Tensor]] = None) -> Dict[str, Dict[str, float]]:
    """`step_logits[k]` = the product's [B, V] logits of output position k, rows in `prompts` order;
    `outs[b]` = the product's tokens.  Returns {flavour: stats, "literal_vs_fp32acc": stats}.
    `forced_topk_ids` ({layer: [B, max_len, top_k]}): mixture-of-experts models evaluated with the routing of the run
    under test (OracleLM.forced_topk_ids), in both flavours."""
    device = device if device is not None else step_logits[0].device
    B, n_new = len(prompts), len(step_logits)
    total = sum(len(p) for p in prompts) + B * n_new + 64
    ctx = max_ctx or (max(len(p) for p in prompts) + n_new + 8)
    w = weights_from_product_model(model, device=device)
    report: Dict[str, Dict[str, float]] = {}
    kept: Dict[str, List[torch.Tensor]] = {}
    for fl in flavours:
        oracle = OracleLM(cfg, w, num_slots=total, max_ctx=ctx, max_reqs=B, device=device,
                          compute_dtype=torch.float32 if fl == "fp32acc" else None, batched_decode=batched_decode)
        oracle.forced_topk_ids = forced_topk_ids
        st = LogitStats(clear_margin_ulps)
        # the second flavour is also compared with the first one: keep the first one's logits in bf16-exact form
        # (they are bf16 values widened to fp32, so the narrow copy loses nothing)
        keep = kept.setdefault(fl, []) if len(flavours) > 1 else None

        def hook(step, ref, st=st, keep=keep):
            st.update(step, step_logits[step].to(ref.device), ref)
            if keep is not None:
                keep.append(ref.to(torch.bfloat16))

        free = oracle.generate(prompts, n_new, forced=outs, logits_hook=hook)
        rep = st.summary()
        # how far the product's greedy run follows the oracle's own choices under teacher forcing
        rep["requests_with_identical_tokens"] = sum(1 for a, b in zip(free, outs) if list(a) == list(b)) / B
        report[fl] = rep
        del oracle
    if len(flavours) == 2:
        a, b = flavours
        st = LogitStats(clear_margin_ulps)
        for k, (x, y) in enumerate(zip(kept[b], kept[a])):
            st.update(k, x, y)
        report[f"{b}_vs_{a}"] = st.summary()
    return report
