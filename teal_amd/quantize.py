"""int8 weight-only quantisation for the sparse decode path.

The reference ships int8 weight-only linears for its DENSE gpt-fast path (gpt-fast/quantize.py:24-56
`dynamically_quantize_per_channel`, :316-337 `WeightOnlyInt8QuantHandler`, :339-357 `WeightOnlyInt8Linear`) and
lists quantisation + TEAL as not yet supported (README.md:110).  Here the same quantiser feeds the HIP sparse
GEMV: kept rows of an int8 W^T are half the bytes of fp16, and the per-column scale is applied to the fp32 sum.

    quantize_per_channel(w)         <- dynamically_quantize_per_channel(w.float(), -128, 127, torch.int8)
    WeightOnlyInt8Linear            <- quantize.py:339-357 (buffers `weight` int8 [N, Z], `scales` [N])
    quantize_model_int8(model)      <- WeightOnlyInt8QuantHandler.create_quantized_state_dict + convert_for_runtime

int4 group-wise (gpt-fast/quantize.py:58-162, 359-443, 483-526), same story: the reference's WeightOnlyInt4Linear runs a
CUDA-only packed matmul on its dense path; here the same group quantiser feeds `teal::sparse_gemv_int4`:

    get_group_qparams / group_quantize_tensor / group_dequantize_tensor   <- quantize.py:58-162 (same arithmetic)
    pack_int4_colmajor(q)           our layout: the nibble image of W^T, [Z][N / 2 + pad] bytes (the reference packs for tinygemm)
    WeightOnlyInt4Linear            buffers `weight` (packed uint8) and `scales_and_zeros` bf16 [Z / G][N][2] (the reference's tensor)
    quantize_model_int4(model, G)   <- WeightOnlyInt4QuantHandler (no padding branch: Llama shapes divide by every G)
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


# ------------------------------------------------------------------------------------------------
# int4 group-wise
# ------------------------------------------------------------------------------------------------
INT4_ROW_PAD_BYTES = 128  # pair-row padding of the packed image: rotates the DRAM channel residue like monkeypatch.ROW_PAD, and keeps
# every 128-byte tile segment inside ONE 128-byte line — a stride of 64 mod 128 made each segment straddle two lines: 538 -> 578 tok/s
# on Llama-2-7B, 89 -> 103 on Llama-2-70B (int4-g32 @ 50 %; 0 / 128 / 384 equal, 256 two per cent behind)


def get_group_qparams(w: torch.Tensor, n_bit: int = 4, groupsize: int = 128):
    """per (output row, group of `groupsize` input features): scale = (max - min) / 15 clamped to >= 1e-6,
    zero = min + scale * 8, both rounded to bf16 (quantize.py:58-76)."""
    assert groupsize > 1 and w.dim() == 2 and w.shape[-1] % groupsize == 0
    g = w.reshape(-1, groupsize)
    assert torch.isnan(g).sum() == 0
    max_val, min_val = g.amax(dim=1, keepdim=True), g.amin(dim=1, keepdim=True)
    max_int = 2 ** n_bit - 1
    scales = (max_val - min_val).clamp(min=1e-6) / max_int
    zeros = min_val + scales * (2 ** (n_bit - 1))
    return scales.to(torch.bfloat16).reshape(w.shape[0], -1), zeros.to(torch.bfloat16).reshape(w.shape[0], -1)


def group_quantize_tensor_from_qparams(w, scales, zeros, n_bit: int = 4, groupsize: int = 128) -> torch.Tensor:
    """q = clamp(round((w - (zero - 8 scale)) / scale), 0, 15) as int32 [N, Z] (quantize.py:101-128)."""
    assert w.dim() == 2 and w.shape[-1] % groupsize == 0
    g = w.reshape(-1, groupsize)
    scales, zeros = scales.reshape(-1, 1), zeros.reshape(-1, 1)
    min_val = zeros - scales * (2 ** (n_bit - 1))
    return g.sub(min_val).div(scales).round().clamp_(0, 2 ** n_bit - 1).to(torch.int32).reshape_as(w)


def group_quantize_tensor(w: torch.Tensor, n_bit: int = 4, groupsize: int = 128):
    """(q int32 [N, Z], scales_and_zeros bf16 [Z / G][N][2]) — quantize.py:131-135 with pack_scales_and_zeros (:79-93)."""
    scales, zeros = get_group_qparams(w, n_bit, groupsize)
    q = group_quantize_tensor_from_qparams(w, scales, zeros, n_bit, groupsize)
    sz = torch.cat([scales.reshape(scales.size(0), scales.size(1), 1), zeros.reshape(zeros.size(0), zeros.size(1), 1)], 2)
    return q, sz.transpose(0, 1).contiguous()


def group_dequantize_tensor(q: torch.Tensor, scales_and_zeros: torch.Tensor, n_bit: int = 4, groupsize: int = 128) -> torch.Tensor:
    """w = (q - 8) * scale + zero, [N, Z] in the dtype of scales_and_zeros (quantize.py:138-162)."""
    scales, zeros = torch.split(scales_and_zeros.transpose(0, 1), 1, 2)
    g = q.reshape(-1, groupsize)
    return g.sub(2 ** (n_bit - 1)).mul(scales.reshape(-1, 1)).add(zeros.reshape(-1, 1)).reshape_as(q)


def pack_int4_colmajor(q: torch.Tensor, pad_bytes: int = INT4_ROW_PAD_BYTES) -> torch.Tensor:
    """q int [N, Z] in 0..15 -> uint8 [Z / 2][N + pad], the image of W^T by row pairs the kernel reads (include/teal_hip.h,
    teal_sparse_qkv_gemv_i4): 32-bit word g of pair-row p = columns 4g .. 4g+3 of row 2p in the low half (nibble j = column
    4g + j) and of row 2p + 1 in the high half, i.e. bytes [2p: c0|c1<<4, 2p: c2|c3<<4, 2p+1: c0|c1<<4, 2p+1: c2|c3<<4]."""
    N, Z = q.shape
    assert N % 4 == 0 and Z % 2 == 0
    qt = q.T.contiguous().to(torch.uint8)  # [Z, N]
    nib = (qt[:, 0::2] | (qt[:, 1::2] << 4)).view(Z // 2, 2, N // 4, 2)  # [pair, row of the pair, word, byte of the half]
    out = torch.zeros(Z // 2, N + pad_bytes, dtype=torch.uint8, device=q.device)
    out[:, :N] = nib.permute(0, 2, 1, 3).reshape(Z // 2, N)
    return out


def unpack_int4_colmajor(packed: torch.Tensor, N: int) -> torch.Tensor:
    """inverse of pack_int4_colmajor -> int32 [N, Z]"""
    P = packed.shape[0]
    nib = packed[:, :N].reshape(P, N // 4, 2, 2).permute(0, 2, 1, 3).reshape(2 * P, N // 2)  # [Z, N / 2]: byte j = columns 2j, 2j+1
    qt = torch.stack((nib & 0xF, nib >> 4), dim=-1).reshape(2 * P, N)
    return qt.T.contiguous().to(torch.int32)


class WeightOnlyInt4Linear(nn.Module):
    """Group-quantised int4 weight-only linear for the sparse decode path.  Dense forward (prefill, any shape) =
    F.linear(x, dequantised weight): what the reference's module computes through its packed matmul
    (quantize.py:483-526), without the CUDA-only packed layout."""

    def __init__(self, in_features: int, out_features: int, groupsize: int = 128, device=None):
        super().__init__()
        assert in_features % groupsize == 0 and out_features % 8 == 0
        self.in_features, self.out_features, self.groupsize = in_features, out_features, groupsize
        self.register_buffer("weight", torch.zeros((in_features // 2, out_features + INT4_ROW_PAD_BYTES), dtype=torch.uint8, device=device))
        self.register_buffer("scales_and_zeros", torch.zeros((in_features // groupsize, out_features, 2), dtype=torch.bfloat16, device=device))

    def dequantized(self, dtype) -> torch.Tensor:
        q = unpack_int4_colmajor(self.weight, self.out_features)
        return group_dequantize_tensor(q, self.scales_and_zeros.float(), 4, self.groupsize).to(dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.linear(x, self.dequantized(x.dtype))

    @classmethod
    def from_linear(cls, lin: nn.Linear, groupsize: int = 128) -> "WeightOnlyInt4Linear":
        assert lin.bias is None
        q, sz = group_quantize_tensor(lin.weight.data.to(torch.bfloat16), 4, groupsize)  # quantize.py:421: quantised from bf16
        m = cls(lin.in_features, lin.out_features, groupsize, device=lin.weight.device)
        m.weight.copy_(pack_int4_colmajor(q))
        m.scales_and_zeros.copy_(sz)
        return m


def int4_kernel_supports(in_features: int, out_features: int, groupsize: int, kv_size: int = 0) -> bool:
    """Shape contract of teal_sparse_qkv_gemv_i4 (include/teal_hip.h): 128-column tiles (N, and for a fused wqkv the q and
    kv widths, multiples of 128), whole groups (Z % G == 0, G in 32 / 64 / 128 / 256), Z <= 65536."""
    return (groupsize in (32, 64, 128, 256) and in_features % groupsize == 0 and in_features <= 65536 and out_features % 128 == 0
            and kv_size % 128 == 0 and (out_features - 2 * kv_size) % 128 == 0 and out_features - 2 * kv_size > 0)


def _int4_eligible(name: str, child: nn.Linear, groupsize: int, kv_size: int) -> bool:
    return int4_kernel_supports(child.in_features, child.out_features, groupsize, kv_size if name == "wqkv" else 0)


def quantize_model_int4(model: nn.Module, groupsize: int = 32, skip=("output",), _kv: int = None) -> nn.Module:
    """Replace the projections' nn.Linear by WeightOnlyInt4Linear in place.  `skip`: child names kept in 16 bits (the
    lm_head: its 32000 / 128256 columns are no multiple of the kernel's 128-column tile for every vocabulary, and the
    reference's handler pads / skips shapes it cannot pack, quantize.py:404-415).  A linear whose shape the sparse int4
    kernel cannot take (teal_sparse_qkv_gemv_i4: 128-column tiles, whole groups) stays in 16 bits too — all five
    projections of a block then do, so that a block is never half quantised — instead of failing at the first decode step."""
    if _kv is None:
        cfg = getattr(model, "config", None)
        _kv = cfg.n_local_heads * cfg.head_dim if cfg is not None else 0
    children = list(model.named_children())
    lin_names = {"wqkv", "wo", "w1", "w2", "w3"}
    for name, child in children:
        if isinstance(child, nn.Linear) and name not in skip:
            setattr(model, name, WeightOnlyInt4Linear.from_linear(child, groupsize))
            del child
        elif isinstance(child, nn.Module) and hasattr(child, "attention") and hasattr(child, "feed_forward"):
            # a transformer block: quantise its five projections together or not at all
            lins = [(n, getattr(sub, n)) for sub in (child.attention, child.feed_forward) for n in lin_names if isinstance(getattr(sub, n, None), nn.Linear)]
            if lins and all(_int4_eligible(n, l, groupsize, _kv) for n, l in lins):
                quantize_model_int4(child, groupsize, skip, _kv)
        else:
            quantize_model_int4(child, groupsize, skip, _kv)
    return model


def convert_for_runtime_int4(model: nn.Module, groupsize: int = 32, skip=("output",), _kv: int = None) -> nn.Module:
    """Replace the projections by EMPTY WeightOnlyInt4Linear modules (role of WeightOnlyInt4QuantHandler.convert_for_runtime,
    quantize.py:417-443): the shape a state dict written by quantize_model_int4(...).state_dict() loads into — packed
    `weight` uint8 [Z / 2][N + 128] (nibble image of W^T by row pairs) and `scales_and_zeros` bf16 [Z / G][N][2].
    NOT interchangeable with the reference's *int4* checkpoints: those hold the CUDA tinygemm tile layout produced by
    aten._convert_weight_to_int4pack (quantize.py:366-372), which only that kernel reads; scales_and_zeros is the same
    tensor in both."""
    if _kv is None:
        cfg = getattr(model, "config", None)
        _kv = cfg.n_local_heads * cfg.head_dim if cfg is not None else 0
    lin_names = {"wqkv", "wo", "w1", "w2", "w3"}
    for name, child in list(model.named_children()):
        if isinstance(child, nn.Linear) and name not in skip:
            setattr(model, name, WeightOnlyInt4Linear(child.in_features, child.out_features, groupsize, device=child.weight.device))
        elif isinstance(child, nn.Module) and hasattr(child, "attention") and hasattr(child, "feed_forward"):
            lins = [(n, getattr(sub, n)) for sub in (child.attention, child.feed_forward) for n in lin_names if isinstance(getattr(sub, n, None), nn.Linear)]
            if lins and all(_int4_eligible(n, l, groupsize, _kv) for n, l in lins):
                convert_for_runtime_int4(child, groupsize, skip, _kv)
        else:
            convert_for_runtime_int4(child, groupsize, skip, _kv)
    return model


def is_int4(lin: nn.Module) -> bool:
    return isinstance(lin, WeightOnlyInt4Linear)


def quantize(checkpoint_path, mode: str = "int8", groupsize: int = 32, label: str = "", device: str = "cpu"):
    """Write the weight-only quantised checkpoint next to `checkpoint_path` — the command-line role of the reference's
    gpt-fast/quantize.py:528-600: model.pth -> model{label}int8.pth / model{label}int4.g{G}.pth, the names the loader's
    branches key on (teal_amd/gpt_fast/generate.py load_checkpoint_model; gpt-fast/generate.py:236-243).  int8 state dicts are
    the reference's (`weight` int8 [N, Z], `scales`); int4 ones hold this build's row-pair image (convert_for_runtime_int4).
    mode 'int4-gptq' (calibration on lm-eval tasks) is outside this build."""
    import time
    from pathlib import Path

    from .gpt_fast.model import Transformer
    checkpoint_path = Path(checkpoint_path)
    assert checkpoint_path.is_file(), checkpoint_path
    if mode not in ("int8", "int4"):
        raise ValueError(f"Invalid quantization mode {mode}: this build writes int8 and int4 (int4-gptq needs the lm-eval calibration "
                         "harness, out of scope)")
    t0 = time.time()
    with torch.device("meta"):
        model = Transformer.from_name(checkpoint_path.parent.name)
    ckpt = torch.load(str(checkpoint_path), mmap=True, weights_only=True)
    model.load_state_dict(ckpt, assign=True)
    model = model.to(dtype=torch.bfloat16, device=device)  # the reference quantises from bf16 (quantize.py:546)
    if mode == "int8":
        print("Quantizing model weights for int8 weight-only symmetric per-channel quantization")
        sd = quantize_model_int8(model).state_dict()
        name = checkpoint_path.name.replace(".pth", f"{label}int8.pth")
    else:
        print("Quantizing model weights for int4 weight-only affine per-channel groupwise quantization")
        sd = quantize_model_int4(model, groupsize).state_dict()
        name = checkpoint_path.name.replace(".pth", f"{label}int4.g{groupsize}.pth")
    out = checkpoint_path.parent / name
    print(f"Writing quantized weights to {out}")
    out.unlink(missing_ok=True)
    torch.save(sd, out)
    print(f"Quantization complete took {time.time() - t0:.02f} seconds")
    return out


if __name__ == "__main__":
    import argparse
    from pathlib import Path
    ap = argparse.ArgumentParser(description="Quantize a model (weight-only int8 / int4 group-wise).")
    ap.add_argument("--checkpoint_path", type=Path, default=Path("checkpoints/meta-llama/Llama-2-7b-chat-hf/model.pth"))
    ap.add_argument("--mode", "-q", type=str, default="int8", choices=["int8", "int4", "int4-gptq"])
    ap.add_argument("--groupsize", type=int, default=32, help="Group size for int4 quantization.")
    ap.add_argument("--label", type=str, default="_", help="label to add to output filename")
    ap.add_argument("--device", type=str, default="cpu", help="where to run the quantiser (cpu works; cuda is faster)")
    a = ap.parse_args()
    quantize(a.checkpoint_path, a.mode, a.groupsize, a.label, a.device)
