"""Model-side plugin swap: what gpt-fast/generate.py:256-331 (`monkeypatch_layer`) does to each
TransformerBlock — load 4 histograms, derive 7 thresholds, install the gemv ops + thresholds as
attributes, re-lay the 5 projection weights column-major, swap the forwards.

The attribute bundle is exactly the reference's:
  feed_forward: gemv1_kernel gemv1 gemv2_kernel gemv2 thresh_up thresh_gate thresh_down sparsity_bin
  attention:    gemv1_kernel gemv1 gemv2_kernel gemv2 thresh_q thresh_k thresh_v thresh_o sparsity_bin
so an unmodified gpt-fast model.py (`_new_attn_forward` / `_new_ffn_forward`, model.py:163-190,
258-259) runs on these ops.  Extensions (not in the reference): per-projection sparsities
(`sparsities` dict — wires the block-wise greedy tables the reference imports but never uses,
generate.py:260-264), and explicit `thresholds` for synthetic models.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import torch

from .distribution import Distribution, threshold_for_sparsity
from .kernels.sparse_gemv import (SparseGEMV, SparseGEMVInt4, SparseGEMVInt8, SparseQKVGEMV, SparseQKVGEMVInt4,
                                  SparseQKVGEMVInt8)
from .utils import PROJS

# projection -> (sub-directory, histogram key)   (gpt-fast/generate.py:278-287)
PROJ_HIST = {"q": ("self_attn", "h1"), "k": ("self_attn", "h1"), "v": ("self_attn", "h1"),
             "o": ("self_attn", "h2"), "gate": ("mlp", "h1"), "up": ("mlp", "h1"), "down": ("mlp", "h2")}


def layer_thresholds(layer_idx: int, hist_path: str, sparsities: Dict[str, Sequence[float]]) -> Dict[str, float]:
    """7 thresholds of one layer from its histograms: tau = icdf(0.5 + 0.5*s)."""
    cache: Dict[tuple, Distribution] = {}
    out = {}
    for proj in PROJS:
        sub, h = PROJ_HIST[proj]
        if (sub, h) not in cache:
            cache[(sub, h)] = Distribution(os.path.join(hist_path, f"layer-{layer_idx}", sub), h)
        out[proj] = threshold_for_sparsity(cache[(sub, h)], sparsities[proj][layer_idx])
    return out


ROW_PAD = 64  # elements (128 B): see to_column_major (round 3 re-check on the decode bench: 64 / 128 / 192 / 320 equal, 256 six per cent slower)


# The up matrix starts this many bytes into its allocation: large allocations are 2 MB-aligned, so gate row m and up row m
# (requested together by the fused gate|up launch) would otherwise sit in the same DRAM channel phase; shifted, a
# Llama-2-7B layer runs 0.3 us faster (53.6 -> 53.3 us, scripts/micro/layer_bench LB_W3OFF, alternating runs on one box).
UP_SHIFT_BYTES = int(os.environ.get("TEAL_UP_SHIFT_BYTES", "1024"))  # the variable exists for the A/B measurement only


def to_column_major(linear: torch.nn.Module, pad: Optional[int] = None, shift_bytes: int = 0) -> None:
    """weight.data = weight.data.T.contiguous().T : shape stays [N, Z], memory becomes W^T [Z][ld]
    (gpt-fast/generate.py:296-317), here with a row stride ld = N + pad.

    Why pad: with ld = N every row of W^T starts at a multiple of 1024 B for all Llama shapes, so a
    workgroup's 128-byte column tile always lands in the same 128-B address residue = the same DRAM
    channel group for every row; on MI355X one such group is ~25 % slower and the workgroups bound to it
    finish last.  ld = N + 64 rotates the residue row by row (measured: qkv 15.8 -> 13.7 us, gate 14.7 ->
    13.0 us per launch, profiles/r01_ld_padding.txt).  The reference's own contract only asks for
    `weight.stride(1) > 1` (kernels/sparse_gemv.py:106), which a padded view satisfies.
    int8 weights (quantize.WeightOnlyInt8Linear) get 128 pad elements: the same 128 bytes."""
    w = linear.weight.data
    N, Z = w.shape
    if pad is None:
        pad = ROW_PAD * 2 // w.element_size()
    ld = N + pad
    if w.stride(0) == 1 and w.stride(1) == ld:
        return
    shift = shift_bytes // w.element_size()  # keeps the 16-byte (and 128-byte tile) alignment of every row
    assert shift_bytes % 128 == 0
    flat = torch.zeros(Z * ld + shift, dtype=w.dtype, device=w.device)
    buf = flat[shift:].view(Z, ld)
    buf[:, :N] = w.T
    linear.weight.data = buf[:, :N].T


def monkeypatch_layer(layer_idx: int, layer, sparsity, hist_path: Optional[str], device: str = "cuda", *,
                      sparsities: Optional[Dict[str, Sequence[float]]] = None,
                      thresholds: Optional[Dict[str, float]] = None) -> Dict[str, float]:
    """Install the sparse-GEMV plugin on one gpt-fast TransformerBlock. Returns the thresholds."""
    if thresholds is None:
        if hist_path is None:
            raise ValueError("need hist_path (calibration histograms) or explicit thresholds")
        if sparsities is None:
            n = layer_idx + 1
            sparsities = {p: [sparsity] * n for p in PROJS}
        thresholds = layer_thresholds(layer_idx, hist_path, sparsities)
    ff, attn = layer.feed_forward, layer.attention
    lins = (ff.w1, ff.w3, ff.w2, attn.wqkv, attn.wo)
    int4 = [hasattr(lin, "scales_and_zeros") for lin in lins]
    if any(int4):
        if not all(int4):
            raise ValueError("either all five projections of a block are int4 group-quantised or none")
        return _monkeypatch_layer_int4(layer, thresholds, device)
    int8 = [lin.weight.dtype == torch.int8 for lin in lins]
    if any(int8):
        if not all(int8):
            raise ValueError("either all five projections of a block are int8 weight-only or none")
        return _monkeypatch_layer_int8(layer, thresholds, device)

    ff.gemv1_kernel = SparseGEMV.initialize("sparse_gemv", device)
    ff.gemv1 = ff.gemv1_kernel.operator(True)
    ff.thresh_up = thresholds["up"]
    ff.thresh_gate = thresholds["gate"]
    ff.sparsity_bin = 0
    to_column_major(ff.w1)
    to_column_major(ff.w3, shift_bytes=UP_SHIFT_BYTES)
    ff.gemv2_kernel = SparseGEMV.initialize("sparse_gemv", device)
    ff.gemv2 = ff.gemv2_kernel.operator(True)
    ff.thresh_down = thresholds["down"]
    to_column_major(ff.w2)

    attn.gemv1_kernel = SparseQKVGEMV.initialize("sparse_qkv_gemv", device)
    attn.gemv1 = attn.gemv1_kernel.operator(True)
    attn.thresh_q = thresholds["q"]
    attn.thresh_k = thresholds["k"]
    attn.thresh_v = thresholds["v"]
    attn.sparsity_bin = 0
    to_column_major(attn.wqkv)
    attn.gemv2_kernel = SparseGEMV.initialize("sparse_gemv", device)
    attn.gemv2 = attn.gemv2_kernel.operator(True)
    attn.thresh_o = thresholds["o"]
    to_column_major(attn.wo)

    ff.apply_monkeypatch()
    attn.apply_monkeypatch()
    if torch.cuda.is_available() and ff.w1.weight.is_cuda:
        torch.cuda.empty_cache()  # release the pre-relayout copies (generate.py:323); matters for 70B in 288 GB
    return thresholds


def _monkeypatch_layer_int8(layer, thresholds: Dict[str, float], device: str) -> Dict[str, float]:
    """Same attribute bundle over int8 weight-only projections (quantize.quantize_model_int8): gemv1 / gemv2 are
    teal::sparse_qkv_gemv_int8 / teal::sparse_gemv_int8, which take the `scales` buffer next to the weight."""
    ff, attn = layer.feed_forward, layer.attention
    ff.gemv1_kernel = SparseGEMVInt8.initialize("sparse_gemv_int8", device)
    ff.gemv1 = ff.gemv1_kernel.operator(True)
    ff.gemv2_kernel = SparseGEMVInt8.initialize("sparse_gemv_int8", device)
    ff.gemv2 = ff.gemv2_kernel.operator(True)
    ff.thresh_up, ff.thresh_gate, ff.thresh_down, ff.sparsity_bin = thresholds["up"], thresholds["gate"], thresholds["down"], 0
    attn.gemv1_kernel = SparseQKVGEMVInt8.initialize("sparse_qkv_gemv_int8", device)
    attn.gemv1 = attn.gemv1_kernel.operator(True)
    attn.gemv2_kernel = SparseGEMVInt8.initialize("sparse_gemv_int8", device)
    attn.gemv2 = attn.gemv2_kernel.operator(True)
    attn.thresh_q, attn.thresh_k, attn.thresh_v, attn.thresh_o = (thresholds[k] for k in ("q", "k", "v", "o"))
    attn.sparsity_bin = 0
    for lin in (ff.w1, ff.w3, ff.w2, attn.wqkv, attn.wo):
        to_column_major(lin, shift_bytes=UP_SHIFT_BYTES if lin is ff.w3 else 0)
    ff.int8 = attn.int8 = True
    ff.apply_monkeypatch()
    attn.apply_monkeypatch()
    if torch.cuda.is_available() and ff.w1.weight.is_cuda:
        torch.cuda.empty_cache()
    return thresholds


def _monkeypatch_layer_int4(layer, thresholds: Dict[str, float], device: str) -> Dict[str, float]:
    """Same attribute bundle over int4 group-quantised projections (quantize.quantize_model_int4): gemv1 / gemv2 are
    teal::sparse_qkv_gemv_int4 / teal::sparse_gemv_int4, which take `scales_and_zeros` next to the packed weight.  The
    packed image is already column-gathered (quantize.pack_int4_colmajor): nothing to re-lay out."""
    ff, attn = layer.feed_forward, layer.attention
    ff.gemv1_kernel = SparseGEMVInt4.initialize("sparse_gemv_int4", device)
    ff.gemv1 = ff.gemv1_kernel.operator(True)
    ff.gemv2_kernel = SparseGEMVInt4.initialize("sparse_gemv_int4", device)
    ff.gemv2 = ff.gemv2_kernel.operator(True)
    ff.thresh_up, ff.thresh_gate, ff.thresh_down, ff.sparsity_bin = thresholds["up"], thresholds["gate"], thresholds["down"], 0
    attn.gemv1_kernel = SparseQKVGEMVInt4.initialize("sparse_qkv_gemv_int4", device)
    attn.gemv1 = attn.gemv1_kernel.operator(True)
    attn.gemv2_kernel = SparseGEMVInt4.initialize("sparse_gemv_int4", device)
    attn.gemv2 = attn.gemv2_kernel.operator(True)
    attn.thresh_q, attn.thresh_k, attn.thresh_v, attn.thresh_o = (thresholds[k] for k in ("q", "k", "v", "o"))
    attn.sparsity_bin = 0
    ff.int4 = attn.int4 = True
    ff.apply_monkeypatch()
    attn.apply_monkeypatch()
    return thresholds
