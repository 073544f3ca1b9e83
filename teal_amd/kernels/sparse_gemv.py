"""Host wrappers of the sparse-GEMV hot path — same names, arguments and error behaviour as the
reference's kernels/sparse_gemv.py, backed by hand-written HIP (libteal_hip.so) instead of Triton.

    splitk_sparse_gemv(x, weight, threshold, sparsity_bin)            <- kernels/sparse_gemv.py:87-142
    qkv_gemv(x, weight, tq, tk, tv, sparsity_bin, kv_size)            <- kernels/sparse_gemv.py:196-237
    SparseGEMV / SparseQKVGEMV / DenseGEMV (torch.library wrappers)   <- kernels/sparse_gemv.py:249-307

Differences, all deliberate:
  * no autotune / no first-call warm-up: launch geometry is a pure function of (Z, N, CU count);
    `sparsity_bin` is accepted and ignored (it only keyed the reference's autotune cache);
  * no separate zero-fill launch, no fp16 atomics: fp32 accumulate, one rounding, deterministic;
  * bf16 is native (the reference computes into an fp16 buffer, prints a warning and casts,
    kernels/sparse_gemv.py:138-140); the result dtype is x.dtype in both cases;
  * no CPU path: CPU tensors raise (the reference would fail inside Triton).
"""
from __future__ import annotations

import torch

from .. import _lib, runtime
from .compile_wrapper import BaseKernel

__all__ = ["splitk_sparse_gemv", "qkv_gemv", "dense_gemv", "sparse_gateup_silu", "compact",
           "SparseGEMV", "SparseQKVGEMV", "DenseGEMV",
           "qkv_gemv_int8", "splitk_sparse_gemv_int8", "SparseGEMVInt8", "SparseQKVGEMVInt8"]


def _prep(x: torch.Tensor, weight: torch.Tensor):
    N, Z = weight.shape
    assert x.shape[2] == Z
    assert weight.stride(1) > 1, "weight should be column major"
    if not x.is_cuda or not weight.is_cuda:
        raise RuntimeError("teal_amd sparse GEMV runs on the GPU only (HIP kernels; there is no CPU fallback)")
    if weight.stride(0) != 1 or weight.stride(1) < N or weight.stride(1) % 8:
        raise RuntimeError("weight must be the reference's column-major layout: weight.T.contiguous().T "
                           "(strides (1, ld) with ld >= N, ld % 8 == 0)")
    if weight.dtype != x.dtype:
        raise TypeError(f"x ({x.dtype}) and weight ({weight.dtype}) must share a dtype")
    x = x.contiguous()
    code = runtime.dtype_code(x.dtype)
    L = _lib.load()
    ws = runtime.workspace(x.device, int(L.teal_workspace_bytes(Z, N)))
    return L, x, N, Z, code, ws


def splitk_sparse_gemv(x: torch.Tensor, weight: torch.Tensor, threshold: float, sparsity_bin: int = 0) -> torch.Tensor:
    """y = sparse(x) @ weight.T for x [1, 1, Z], weight [N, Z] column-major; rows with
    float32(|x|) <= float32(threshold) are never read from HBM."""
    L, x, N, Z, code, ws = _prep(x, weight)
    B, S, _ = x.shape
    if B * S != 1:
        raise RuntimeError("splitk_sparse_gemv is the single-token path: x must be [1, 1, Z] "
                           "(the reference kernel only implements BATCHSIZE == 1)")
    y = torch.empty(B, S, N, device=x.device, dtype=x.dtype)
    t = float(threshold)
    rc = L.teal_sparse_qkv_gemv_ld(x.data_ptr(), weight.data_ptr(), weight.stride(1), y.data_ptr(), t, t, t, Z, N, N, 0,
                                   code, ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
    _lib.check(rc, "teal_sparse_gemv")
    return y


def qkv_gemv(x: torch.Tensor, weight: torch.Tensor, threshold_q: float, threshold_k: float, threshold_v: float,
             sparsity_bin: int, kv_size: int) -> torch.Tensor:
    """Fused wqkv projection with one threshold per q / k / v column range."""
    L, x, N, Z, code, ws = _prep(x, weight)
    B, S, _ = x.shape
    if B * S != 1:
        raise RuntimeError("qkv_gemv is the single-token path: x must be [1, 1, Z]")
    N_q = N - 2 * kv_size
    y = torch.empty(B, S, N, device=x.device, dtype=x.dtype)
    rc = L.teal_sparse_qkv_gemv_ld(x.data_ptr(), weight.data_ptr(), weight.stride(1), y.data_ptr(), float(threshold_q),
                                   float(threshold_k), float(threshold_v), Z, N, N_q, kv_size, code,
                                   ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
    _lib.check(rc, "teal_sparse_qkv_gemv")
    return y


def dense_gemv(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """x @ weight.T at one token with every row kept (same kernel, threshold -inf)."""
    L, x, N, Z, code, ws = _prep(x, weight)
    B, S, _ = x.shape
    if B * S != 1:
        return torch.matmul(x, weight.T)
    y = torch.empty(B, S, N, device=x.device, dtype=x.dtype)
    ninf = float("-inf")
    rc = L.teal_sparse_qkv_gemv_ld(x.data_ptr(), weight.data_ptr(), weight.stride(1), y.data_ptr(), ninf, ninf, ninf, Z, N, N, 0,
                                   code, ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
    _lib.check(rc, "teal_dense_gemv")
    return y


def sparse_gateup_silu(x: torch.Tensor, w1: torch.Tensor, w3: torch.Tensor, threshold_gate: float,
                       threshold_up: float) -> torch.Tensor:
    """silu(gemv(x, w1, tau_gate)) * gemv(x, w3, tau_up) in one GEMV launch + one epilogue
    (fusion of gpt-fast/model.py:258-259's two gemv1 calls and the activation)."""
    L, x, N, Z, code, ws = _prep(x, w1)
    if w3.shape != w1.shape or w3.stride() != w1.stride() or w3.dtype != w1.dtype:
        raise RuntimeError("w1 and w3 must have identical shape, layout and dtype")
    if w1.stride(1) != N:
        raise RuntimeError("sparse_gateup_silu takes unpadded weights (strides (1, N)); the engine handles padded rows")
    B, S, _ = x.shape
    if B * S != 1:
        raise RuntimeError("sparse_gateup_silu is the single-token path: x must be [1, 1, Z]")
    h = torch.empty(B, S, N, device=x.device, dtype=x.dtype)
    rc = L.teal_sparse_gateup_silu(x.data_ptr(), w1.data_ptr(), w3.data_ptr(), h.data_ptr(), float(threshold_gate),
                                   float(threshold_up), Z, N, code, ws.data_ptr(), ws.numel() * 4,
                                   runtime.stream_ptr())
    _lib.check(rc, "teal_sparse_gateup_silu")
    return h


def qkv_gemv_int8(x: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor, threshold_q: float, threshold_k: float,
                  threshold_v: float, sparsity_bin: int, kv_size: int) -> torch.Tensor:
    """qkv_gemv on int8 weight-only-quantised weights: weight int8 [N, Z] column-major
    (strides (1, ld)), scales [N] in x.dtype — the buffers of the reference's WeightOnlyInt8Linear
    (gpt-fast/quantize.py:339-357), re-laid like the fp16 path.  y = (sparse(x) @ weight.T) * scales with one
    rounding.  kv_size = 0: a single threshold (threshold_q)."""
    N, Z = weight.shape
    assert x.shape[2] == Z
    assert weight.stride(1) > 1, "weight should be column major"
    if not x.is_cuda or not weight.is_cuda or not scales.is_cuda:
        raise RuntimeError("teal_amd sparse GEMV runs on the GPU only (HIP kernels; there is no CPU fallback)")
    if weight.dtype != torch.int8 or weight.stride(0) != 1 or weight.stride(1) < N or weight.stride(1) % 8:
        raise RuntimeError("int8 weight must be column-major int8: strides (1, ld) with ld >= N, ld % 8 == 0")
    if scales.dtype != x.dtype or scales.numel() != N or not scales.is_contiguous():
        raise TypeError("scales must be a contiguous [N] tensor in x.dtype")
    B, S, _ = x.shape
    if B * S != 1:
        raise RuntimeError("qkv_gemv_int8 is the single-token path: x must be [1, 1, Z]")
    x = x.contiguous()
    L = _lib.load()
    ws = runtime.workspace(x.device, int(L.teal_workspace_bytes(Z, N)))
    y = torch.empty(B, S, N, device=x.device, dtype=x.dtype)
    rc = L.teal_sparse_qkv_gemv_i8(x.data_ptr(), weight.data_ptr(), scales.data_ptr(), y.data_ptr(), float(threshold_q),
                                   float(threshold_k), float(threshold_v), Z, N, N - 2 * kv_size, kv_size, weight.stride(1),
                                   runtime.dtype_code(x.dtype), ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
    _lib.check(rc, "teal_sparse_qkv_gemv_i8")
    return y


def splitk_sparse_gemv_int8(x: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor, threshold: float,
                            sparsity_bin: int = 0) -> torch.Tensor:
    t = float(threshold)
    return qkv_gemv_int8(x, weight, scales, t, t, t, sparsity_bin, 0)


def qkv_gemv_int4(x: torch.Tensor, weight: torch.Tensor, scales_and_zeros: torch.Tensor, threshold_q: float, threshold_k: float,
                  threshold_v: float, sparsity_bin: int, kv_size: int) -> torch.Tensor:
    """qkv_gemv on int4 group-quantised weights: weight = packed nibble image of W^T by row
    pairs, uint8 [Z / 2][N + pad] (quantize.pack_int4_colmajor); scales_and_zeros bf16 [Z / G][N][2] (the reference's tensor,
    gpt-fast/quantize.py:79-93).  y = sparse(x) @ dequant(W).T, fp32 accumulation, one rounding.  kv_size = 0: one threshold."""
    if not x.is_cuda or not weight.is_cuda or not scales_and_zeros.is_cuda:
        raise RuntimeError("teal_amd sparse GEMV runs on the GPU only (HIP kernels; there is no CPU fallback)")
    if weight.dtype != torch.uint8 or weight.dim() != 2 or weight.stride(1) != 1:
        raise RuntimeError("int4 weight must be the packed uint8 image [Z / 2][bytes] (quantize.pack_int4_colmajor)")
    if scales_and_zeros.dtype != torch.bfloat16 or scales_and_zeros.dim() != 3 or scales_and_zeros.shape[2] != 2 or not scales_and_zeros.is_contiguous():
        raise TypeError("scales_and_zeros must be a contiguous bf16 [Z / G][N][2] tensor")
    Z, N = 2 * weight.shape[0], scales_and_zeros.shape[1]
    G = Z // scales_and_zeros.shape[0]
    assert x.shape[2] == Z and scales_and_zeros.shape[0] * G == Z
    B, S, _ = x.shape
    if B * S != 1:
        raise RuntimeError("qkv_gemv_int4 is the single-token path: x must be [1, 1, Z]")
    x = x.contiguous()
    L = _lib.load()
    ws = runtime.workspace(x.device, int(L.teal_workspace_bytes(Z, N)))
    y = torch.empty(B, S, N, device=x.device, dtype=x.dtype)
    rc = L.teal_sparse_qkv_gemv_i4(x.data_ptr(), weight.data_ptr(), scales_and_zeros.data_ptr(), y.data_ptr(), float(threshold_q),
                                   float(threshold_k), float(threshold_v), Z, N, N - 2 * kv_size, kv_size, weight.stride(0), G,
                                   runtime.dtype_code(x.dtype), ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
    _lib.check(rc, "teal_sparse_qkv_gemv_i4")
    return y


def splitk_sparse_gemv_int4(x: torch.Tensor, weight: torch.Tensor, scales_and_zeros: torch.Tensor, threshold: float,
                            sparsity_bin: int = 0) -> torch.Tensor:
    t = float(threshold)
    return qkv_gemv_int4(x, weight, scales_and_zeros, t, t, t, sparsity_bin, 0)


def compact(x: torch.Tensor, threshold: float):
    """(ascending kept indices int32 [count], count) of float32(|x|) > float32(threshold) — the
    index set the GEMV consumes, exposed for bit-exact parity tests."""
    if not x.is_cuda:
        raise RuntimeError("teal_amd.compact runs on the GPU only")
    runtime.init()
    L = _lib.load()
    xf = x.contiguous().view(-1)
    Z = xf.numel()
    idx = torch.empty(Z, dtype=torch.int32, device=x.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=x.device)
    rc = L.teal_compact(xf.data_ptr(), float(threshold), Z, runtime.dtype_code(x.dtype), idx.data_ptr(),
                        cnt.data_ptr(), runtime.stream_ptr())
    _lib.check(rc, "teal_compact")
    n = int(cnt.item())
    return idx[:n], n


# ---- torch.library wrappers (what monkeypatch_layer installs as gemv1 / gemv2) -------------------
class SparseGEMV(BaseKernel):
    def meta(self, hidden_states: torch.Tensor, weights: torch.Tensor, threshold: float,
             sparsity_bin: int) -> torch.Tensor:
        return hidden_states.new_empty((hidden_states.size(0), hidden_states.size(1), weights.size(0)))

    def forward(self, hidden_states: torch.Tensor, weights: torch.Tensor, threshold: float,
                sparsity_bin: int) -> torch.Tensor:
        # decode -> HIP sparse GEMV; prefill -> dense matmul (kernels/sparse_gemv.py:271)
        if hidden_states.shape[1] == 1 and hidden_states.shape[0] == 1:
            return splitk_sparse_gemv(hidden_states, weights, threshold, sparsity_bin)
        return torch.matmul(hidden_states, weights.T)


class SparseQKVGEMV(BaseKernel):
    def meta(self, x: torch.Tensor, weight: torch.Tensor, threshold_q: float, threshold_k: float,
             threshold_v: float, sparsity_bin: int, kv_size: int) -> torch.Tensor:
        return x.new_empty(x.shape[0], x.shape[1], weight.shape[0])

    def forward(self, x: torch.Tensor, weight: torch.Tensor, threshold_q: float, threshold_k: float,
                threshold_v: float, sparsity_bin: int, kv_size: int) -> torch.Tensor:
        if x.shape[1] == 1 and x.shape[0] == 1:
            return qkv_gemv(x, weight, threshold_q, threshold_k, threshold_v, sparsity_bin, kv_size)
        return torch.matmul(x, weight.T)


class DenseGEMV(BaseKernel):
    """Dense comparator with the sparse ops' call shape (reference: kernels/sparse_gemv.py:301-307,
    whose *args/**kwargs signature cannot be schematized; here the schema is explicit)."""

    def meta(self, x: torch.Tensor, W: torch.Tensor, threshold: float, sparsity_bin: int) -> torch.Tensor:
        return x.new_empty(x.shape[0], x.shape[1], W.shape[0])

    def forward(self, x: torch.Tensor, W: torch.Tensor, threshold: float, sparsity_bin: int) -> torch.Tensor:
        if x.shape[1] == 1 and x.shape[0] == 1:
            return dense_gemv(x, W)
        return torch.matmul(x, W.T)


def _int8_prefill(x: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    # WeightOnlyInt8Linear.forward (gpt-fast/quantize.py:354)
    return torch.matmul(x, weight.to(dtype=x.dtype).T) * scales


class SparseGEMVInt8(BaseKernel):
    def meta(self, hidden_states: torch.Tensor, weights: torch.Tensor, scales: torch.Tensor, threshold: float,
             sparsity_bin: int) -> torch.Tensor:
        return hidden_states.new_empty((hidden_states.size(0), hidden_states.size(1), weights.size(0)))

    def forward(self, hidden_states: torch.Tensor, weights: torch.Tensor, scales: torch.Tensor, threshold: float,
                sparsity_bin: int) -> torch.Tensor:
        if hidden_states.shape[1] == 1 and hidden_states.shape[0] == 1:
            return splitk_sparse_gemv_int8(hidden_states, weights, scales, threshold, sparsity_bin)
        return _int8_prefill(hidden_states, weights, scales)


class SparseQKVGEMVInt8(BaseKernel):
    def meta(self, x: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor, threshold_q: float, threshold_k: float,
             threshold_v: float, sparsity_bin: int, kv_size: int) -> torch.Tensor:
        return x.new_empty(x.shape[0], x.shape[1], weight.shape[0])

    def forward(self, x: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor, threshold_q: float, threshold_k: float,
                threshold_v: float, sparsity_bin: int, kv_size: int) -> torch.Tensor:
        if x.shape[1] == 1 and x.shape[0] == 1:
            return qkv_gemv_int8(x, weight, scales, threshold_q, threshold_k, threshold_v, sparsity_bin, kv_size)
        return _int8_prefill(x, weight, scales)


def _int4_prefill(x: torch.Tensor, weight: torch.Tensor, scales_and_zeros: torch.Tensor) -> torch.Tensor:
    # F.linear on the dequantised weight (quantize.WeightOnlyInt4Linear.forward)
    from ..quantize import group_dequantize_tensor, unpack_int4_colmajor
    N = scales_and_zeros.shape[1]
    G = 2 * weight.shape[0] // scales_and_zeros.shape[0]  # the image holds row pairs
    w = group_dequantize_tensor(unpack_int4_colmajor(weight, N), scales_and_zeros.float(), 4, G).to(x.dtype)
    return torch.matmul(x, w.T)


class SparseGEMVInt4(BaseKernel):
    def meta(self, hidden_states: torch.Tensor, weights: torch.Tensor, scales_and_zeros: torch.Tensor, threshold: float,
             sparsity_bin: int) -> torch.Tensor:
        return hidden_states.new_empty((hidden_states.size(0), hidden_states.size(1), scales_and_zeros.size(1)))

    def forward(self, hidden_states: torch.Tensor, weights: torch.Tensor, scales_and_zeros: torch.Tensor, threshold: float,
                sparsity_bin: int) -> torch.Tensor:
        if hidden_states.shape[1] == 1 and hidden_states.shape[0] == 1:
            return splitk_sparse_gemv_int4(hidden_states, weights, scales_and_zeros, threshold, sparsity_bin)
        return _int4_prefill(hidden_states, weights, scales_and_zeros)


class SparseQKVGEMVInt4(BaseKernel):
    def meta(self, x: torch.Tensor, weight: torch.Tensor, scales_and_zeros: torch.Tensor, threshold_q: float, threshold_k: float,
             threshold_v: float, sparsity_bin: int, kv_size: int) -> torch.Tensor:
        return x.new_empty(x.shape[0], x.shape[1], scales_and_zeros.shape[1])

    def forward(self, x: torch.Tensor, weight: torch.Tensor, scales_and_zeros: torch.Tensor, threshold_q: float, threshold_k: float,
                threshold_v: float, sparsity_bin: int, kv_size: int) -> torch.Tensor:
        if x.shape[1] == 1 and x.shape[0] == 1:
            return qkv_gemv_int4(x, weight, scales_and_zeros, threshold_q, threshold_k, threshold_v, sparsity_bin, kv_size)
        return _int4_prefill(x, weight, scales_and_zeros)
