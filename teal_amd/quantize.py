"""int8 weight-only quantisation for the sparse decode path.

The reference ships int8 weight-only linears for its DENSE gpt-fast path (gpt-fast/quantize.py:24-56
`dynamically_quantize_per_channel`, :316-337 `WeightOnlyInt8QuantHandler`, :339-357 `WeightOnlyInt8Linear`) and
lists quantisation + TEAL as not yet supported (README.md:110).  Here the same quantiser feeds the HIP sparse
GEMV: kept rows of an int8 W^T are half the bytes of fp16, and the per-column scale is applied to the fp32 sum.

    quantize_per_channel(w)         <- dynamically_quantize_per_channel(w.float(), -128, 127, torch.int8)
    WeightOnlyInt8Linear            <- quantize.py:339-357 (buffers `weight` int8 [N, Z], `scales` [N])
    quantize_model_int8(model)      <- WeightOnlyInt8QuantHandler.create_quantized_state_dict + convert_for_runtime
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def quantize_per_channel(w: torch.Tensor, quant_min: int = -128, quant_max: int = 127):
    """Symmetric per-output-channel quantisation of w [N, Z] (quantize.py:24-56): scale = max(|row|) / 127.5
    clamped to >= eps(fp32), q = clamp(round(w / scale), -128, 127).  Returns (int8 [N, Z], fp32 scales [N])."""
    x = w.float()
    eps = torch.finfo(torch.float32).eps
    min_val, max_val = torch.aminmax(x, dim=1)
    min_neg = torch.minimum(min_val, torch.zeros_like(min_val))
    max_pos = torch.maximum(max_val, torch.zeros_like(max_val))
    amax = torch.maximum(-min_neg, max_pos)
    scales = torch.clamp(amax / (float(quant_max - quant_min) / 2), min=eps)
    q = torch.clamp(torch.round(x / scales.unsqueeze(-1)), quant_min, quant_max).to(torch.int8)
    return q, scales


class WeightOnlyInt8Linear(nn.Module):
    """Same buffers and dense forward as the reference module (quantize.py:339-357); `weight` may be re-laid
    column-major (strides (1, ld)) by monkeypatch.to_column_major for the HIP path — the values are unchanged."""

    def __init__(self, in_features: int, out_features: int, device=None, dtype=torch.bfloat16):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.register_buffer("weight", torch.empty((out_features, in_features), dtype=torch.int8, device=device))
        self.register_buffer("scales", torch.ones(out_features, dtype=dtype, device=device))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.linear(x, self.weight.to(dtype=x.dtype)) * self.scales

    @classmethod
    def from_linear(cls, lin: nn.Linear) -> "WeightOnlyInt8Linear":
        assert lin.bias is None
        q, s = quantize_per_channel(lin.weight.data)
        m = cls(lin.in_features, lin.out_features, device=lin.weight.device, dtype=lin.weight.dtype)
        m.weight.copy_(q)
        m.scales.copy_(s.to(lin.weight.dtype))  # quantize.py:330: scales stored in the model dtype
        return m


def quantize_model_int8(model: nn.Module) -> nn.Module:
    """Replace every nn.Linear (projections AND lm_head, as the reference handler does) in place."""
    for name, child in list(model.named_children()):
        if isinstance(child, nn.Linear):
            setattr(model, name, WeightOnlyInt8Linear.from_linear(child))
            del child
        else:
            quantize_model_int8(child)
    return model


def convert_for_runtime_int8(model: nn.Module, dtype=torch.bfloat16) -> nn.Module:
    """Replace every nn.Linear by an EMPTY WeightOnlyInt8Linear (WeightOnlyInt8QuantHandler.convert_for_runtime,
    quantize.py:332-337 / replace_linear_weight_only_int8_per_channel): the shape an int8 checkpoint's state dict
    (`weight` int8, `scales`) loads into."""
    for name, child in list(model.named_children()):
        if isinstance(child, nn.Linear):
            dev = child.weight.device
            setattr(model, name, WeightOnlyInt8Linear(child.in_features, child.out_features, device=dev, dtype=dtype))
        else:
            convert_for_runtime_int8(child, dtype)
    return model


def is_int8(lin: nn.Module) -> bool:
    return isinstance(lin, WeightOnlyInt8Linear)
