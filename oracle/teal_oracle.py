"""ctypes front-end of the CPU oracle (oracle/teal_oracle.c) + numpy restatements.

TEST INFRASTRUCTURE ONLY — see the header of oracle/teal_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package teal_amd/ never does.

Reference anchors (into /root/reference):
  keep rule .................. kernels/sparse_gemv.py:75
  split-K sparse GEMV ........ kernels/sparse_gemv.py:50-83
  3-threshold QKV GEMV ....... kernels/sparse_gemv.py:152-194
  SparsifyFn.apply ........... utils/utils.py:51-52
  Distribution.icdf .......... gpt-fast/distribution.py:48-66
  greedy lookup .............. utils/utils.py:243-259
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libteal_oracle.so")
_lib = None

F16, BF16 = 0, 1


def build(force: bool = False) -> str:
    """Compile oracle/teal_oracle.c with gcc (Makefile next to it)."""
    src = os.path.join(_HERE, "teal_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libteal_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u16p = ctypes.POINTER(ctypes.c_uint16)
        i32p = ctypes.POINTER(ctypes.c_int32)
        f64p = ctypes.POINTER(ctypes.c_double)
        c_int, c_float = ctypes.c_int, ctypes.c_float
        L.teal_oracle_compact.argtypes = [u16p, c_int, c_int, c_float, i32p, i32p]
        L.teal_oracle_sparsify_fn_apply.argtypes = [u16p, c_int, c_int, ctypes.c_uint16, u16p]
        L.teal_oracle_ref_qkv_gemv.argtypes = [u16p, u16p, u16p, c_float, c_float, c_float, c_int, c_int,
                                               c_int, c_int, c_int, c_int, c_int]
        L.teal_oracle_ref_sparse_gemv.argtypes = [u16p, u16p, u16p, c_float, c_int, c_int, c_int, c_int, c_int]
        L.teal_oracle_truth64.argtypes = [u16p, u16p, f64p, c_float, c_float, c_float, c_int, c_int, c_int,
                                          c_int, c_int]
        L.teal_oracle_fast_qkv_gemv.argtypes = [u16p, u16p, u16p, c_float, c_float, c_float, c_int, c_int,
                                                c_int, c_int, c_int]
        L.teal_oracle_fast_sparse_gemv.argtypes = [u16p, u16p, u16p, c_float, c_int, c_int, c_int]
        L.teal_oracle_fast_dense_gemv.argtypes = [u16p, u16p, u16p, c_int, c_int, c_int]
        L.teal_oracle_mat_create.argtypes = [u16p, c_int, c_int, c_int]
        L.teal_oracle_mat_create.restype = ctypes.c_void_p
        L.teal_oracle_mat_gemv.argtypes = [ctypes.c_void_p, u16p, u16p, c_float]
        L.teal_oracle_mat_free.argtypes = [ctypes.c_void_p]
        L.teal_oracle_mat_free.restype = None
        L.teal_oracle_host_read_gbs.argtypes = [ctypes.c_size_t, c_int]
        L.teal_oracle_host_read_gbs.restype = ctypes.c_double
        L.teal_oracle_hash_uniform.argtypes = [u16p, ctypes.c_size_t, ctypes.c_uint32, c_float, c_int]
        L.teal_oracle_num_threads.restype = c_int
        L.teal_oracle_set_threads.argtypes = [c_int]
        L.teal_oracle_half_to_float.argtypes = [ctypes.c_uint16]
        L.teal_oracle_half_to_float.restype = c_float
        L.teal_oracle_float_to_half.argtypes = [c_float]
        L.teal_oracle_float_to_half.restype = ctypes.c_uint16
        L.teal_oracle_bf16_to_float.argtypes = [ctypes.c_uint16]
        L.teal_oracle_bf16_to_float.restype = c_float
        L.teal_oracle_float_to_bf16.argtypes = [c_float]
        L.teal_oracle_float_to_bf16.restype = ctypes.c_uint16
        _lib = L
    return _lib


def _u16(a: np.ndarray):
    assert a.dtype == np.uint16 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


# ----------------------------------------------------------------------------------------
# raw-bits helpers: all oracle entry points take uint16 bit patterns
# ----------------------------------------------------------------------------------------
def to_bits(a, dtype: int) -> np.ndarray:
    """float array -> raw 16-bit patterns of fp16 (dtype 0) or bf16 (dtype 1), RNE."""
    a = np.asarray(a)
    if a.dtype == np.uint16:
        return np.ascontiguousarray(a)
    if dtype == F16:
        return np.ascontiguousarray(a.astype(np.float16)).view(np.uint16)
    f = np.ascontiguousarray(a.astype(np.float32)).view(np.uint32)
    lsb = (f >> 16) & 1
    r = ((f + 0x7FFF + lsb) >> 16).astype(np.uint16)
    nan = (f & 0x7FFFFFFF) > 0x7F800000
    r[nan] = ((f[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def from_bits(b: np.ndarray, dtype: int) -> np.ndarray:
    """raw 16-bit patterns -> float32 values."""
    b = np.ascontiguousarray(b).view(np.uint16)
    if dtype == F16:
        return b.view(np.float16).astype(np.float32)
    return (b.astype(np.uint32) << 16).view(np.float32)


def ulp16(v, dtype: int) -> np.ndarray:
    """spacing of the 16-bit format at |v| (fp16: 10 mantissa bits, bf16: 7)."""
    v = np.abs(np.asarray(v, dtype=np.float64))
    mant = 10 if dtype == F16 else 7
    emin = -14 if dtype == F16 else -126
    e = np.floor(np.log2(np.maximum(v, 2.0 ** emin)))
    return 2.0 ** (e - mant)


# ----------------------------------------------------------------------------------------
# portable data generator (bit-identical to teal_oracle_hash_uniform in C)
# ----------------------------------------------------------------------------------------
def hash_uniform(n: int, seed: int, scale: float = 1.0, dtype: int = F16, offset: int = 0) -> np.ndarray:
    """U(-0.5, 0.5)*scale on an 11-bit grid, as raw 16-bit patterns. numpy restatement."""
    i = (np.arange(offset, offset + n, dtype=np.uint64)) & 0xFFFFFFFF
    h = (i * np.uint64(2654435761) + np.uint64((seed * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & 0xFFFFFFFF
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & 0xFFFFFFFF
    h ^= h >> np.uint64(16)
    v = ((h >> np.uint64(21)).astype(np.float32) - np.float32(1024.0)) * np.float32(1.0 / 2048.0) * np.float32(scale)
    return to_bits(v, dtype)


def hash_uniform_c(n: int, seed: int, scale: float = 1.0, dtype: int = F16) -> np.ndarray:
    out = np.empty(n, dtype=np.uint16)
    _check(lib().teal_oracle_hash_uniform(_u16(out), n, seed & 0xFFFFFFFF, scale, dtype), "hash_uniform")
    return out


# ----------------------------------------------------------------------------------------
# keep rule / compaction
# ----------------------------------------------------------------------------------------
def compact(x_bits: np.ndarray, tau: float, dtype: int = F16) -> np.ndarray:
    """ascending indices m with float32(|x[m]|) > float32(tau) (kernels/sparse_gemv.py:75)."""
    x_bits = np.ascontiguousarray(x_bits.reshape(-1))
    idx = np.empty(x_bits.size, dtype=np.int32)
    cnt = ctypes.c_int32(0)
    _check(lib().teal_oracle_compact(_u16(x_bits), dtype, x_bits.size, tau,
                                     idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(cnt)),
           "compact")
    return idx[: cnt.value].copy()


def compact_np(x_bits: np.ndarray, tau: float, dtype: int = F16) -> np.ndarray:
    """independent numpy restatement of the same rule."""
    v = from_bits(x_bits.reshape(-1), dtype)
    return np.nonzero(np.abs(v) > np.float32(tau))[0].astype(np.int32)


def sparsify_fn_apply(x_bits: np.ndarray, tau: float, dtype: int = F16) -> np.ndarray:
    """utils/utils.py:51-52 semantics (threshold rounded to x's dtype before the compare)."""
    x_bits = np.ascontiguousarray(x_bits.reshape(-1))
    out = np.empty_like(x_bits)
    t = int(to_bits(np.array([tau], dtype=np.float32), dtype)[0])
    _check(lib().teal_oracle_sparsify_fn_apply(_u16(x_bits), dtype, x_bits.size, t, _u16(out)), "sparsify")
    return out


# ----------------------------------------------------------------------------------------
# GEMV restatements
# ----------------------------------------------------------------------------------------
def ref_sparse_gemv(x_bits, wT_bits, tau, Z, N, dtype=F16, block_m=128, block_n=512) -> np.ndarray:
    """bit-level restatement of splitk_sparse_gemv_kernel (fp16 output bits)."""
    y = np.empty(N, dtype=np.uint16)
    _check(lib().teal_oracle_ref_sparse_gemv(_u16(x_bits), _u16(wT_bits), _u16(y), tau, Z, N, dtype,
                                             block_m, block_n), "ref_sparse_gemv")
    return y


def ref_qkv_gemv(x_bits, wT_bits, tq, tk, tv, Z, N, N_q, N_kv, dtype=F16, block_m=128, block_n=512):
    y = np.empty(N, dtype=np.uint16)
    _check(lib().teal_oracle_ref_qkv_gemv(_u16(x_bits), _u16(wT_bits), _u16(y), tq, tk, tv, Z, N, N_q, N_kv,
                                          dtype, block_m, block_n), "ref_qkv_gemv")
    return y


def truth64(x_bits, wT_bits, Z, N, tq, tk=None, tv=None, N_q=None, N_kv=0, dtype=F16) -> np.ndarray:
    """double-precision sum over kept rows (thresholds per column range)."""
    if tk is None:
        tk, tv, N_q, N_kv = tq, tq, N, 0
    y = np.empty(N, dtype=np.float64)
    _check(lib().teal_oracle_truth64(_u16(x_bits), _u16(wT_bits), y.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                     tq, tk, tv, Z, N, N_q, N_kv, dtype), "truth64")
    return y


def truth64_np(x_bits, wT_bits, Z, N, tau, dtype=F16) -> np.ndarray:
    """independent numpy restatement (small shapes)."""
    x = from_bits(x_bits, dtype).astype(np.float64)
    W = from_bits(wT_bits, dtype).astype(np.float64).reshape(Z, N)
    keep = np.abs(from_bits(x_bits, dtype)) > np.float32(tau)
    return (W[keep] * x[keep, None]).sum(axis=0)


def pick_threads_resident(x_bits, wT_bits, tau, Z, N, dtype=F16, candidates=(8, 16, 32, 64, 96, 128, 192, 256)) -> int:
    """OpenMP width for the resident-matrix baseline: the fastest of the candidate widths on one GEMV (a Mat is bound to
    the width it was created under, so each candidate prepares its own)."""
    import os
    import time
    best, best_t = 1, float("inf")
    ncpu = os.cpu_count() or 1
    for n in sorted(set(c for c in candidates if c <= ncpu) | {min(ncpu, 8)}):
        set_threads(n)
        m = Mat(wT_bits, Z, N, dtype)
        m.gemv(x_bits, tau)
        t0 = time.perf_counter()
        for _ in range(3):
            m.gemv(x_bits, tau)
        t = (time.perf_counter() - t0) / 3
        m.close()
        if t < best_t:
            best, best_t = n, t
    set_threads(best)
    return best


# ----------------------------------------------------------------------------------------
# int8 weight-only quantisation (numpy; gpt-fast/quantize.py) — pinned by tests/golden/kat_int8.npz
# ----------------------------------------------------------------------------------------
def quantize_per_channel_np(w: np.ndarray):
    """dynamically_quantize_per_channel(w.float(), -128, 127, int8) (quantize.py:24-56): per-row symmetric scale
    = max(-min(row, 0), max(row, 0)) / 127.5 clamped to >= eps(fp32); q = clamp(round_half_even(w / scale)).
    All in float32 like the reference.  Returns (int8 [N, Z], float32 scales [N])."""
    x = np.asarray(w, dtype=np.float32)
    eps = np.finfo(np.float32).eps
    mn = np.minimum(x.min(axis=1), np.float32(0))
    mx = np.maximum(x.max(axis=1), np.float32(0))
    amax = np.maximum(-mn, mx).astype(np.float32)
    scales = np.maximum(amax / np.float32(127.5), np.float32(eps)).astype(np.float32)
    q = np.clip(np.round((x / scales[:, None]).astype(np.float32)), -128, 127).astype(np.int8)
    return q, scales


def int8_keep(x_bits, taus, dtype=F16):
    """keep masks of the kernel rule per threshold (kernels/sparse_gemv.py:75)."""
    x = from_bits(x_bits, dtype)
    return [np.abs(x) > np.float32(t) for t in taus]


def int8_truth64(x_bits, q, scale_bits, tq, tk=None, tv=None, N_q=None, N_kv=0, dtype=F16) -> np.ndarray:
    """double-precision y[n] = scale[n] * sum_{m kept for n's column range} q[n, m] * x[m]  (q: int8 [N, Z])."""
    N, Z = q.shape
    if tk is None:
        tk, tv, N_q, N_kv = tq, tq, N, 0
    x = from_bits(x_bits, dtype).astype(np.float64)
    sc = from_bits(scale_bits, dtype).astype(np.float64)
    kq, kk, kv = int8_keep(x_bits, (tq, tk, tv), dtype)
    y = np.empty(N, dtype=np.float64)
    Q = q.astype(np.float64)
    for lo, hi, k in ((0, N_q, kq), (N_q, N_q + N_kv, kk), (N_q + N_kv, N, kv)):
        if hi > lo:
            y[lo:hi] = (Q[lo:hi][:, k] @ x[k]) * sc[lo:hi]
    return y


def int8_ref_forward(x_bits, q, scale_bits, tau, dtype=F16) -> np.ndarray:
    """WeightOnlyInt8Linear.forward (quantize.py:354) on the TEAL-masked activation, with the reference's two
    roundings: y16 = round(F.linear(x_masked, q.to(dtype))), out = round(y16 * scale).  Returns 16-bit patterns."""
    x = from_bits(x_bits, dtype).astype(np.float64)
    keep = int8_keep(x_bits, (tau,), dtype)[0]
    acc = q.astype(np.float64)[:, keep] @ x[keep]
    y16 = from_bits(to_bits(acc.astype(np.float32), dtype), dtype)
    return to_bits(y16 * from_bits(scale_bits, dtype), dtype)


def fast_sparse_gemv(x_bits, wT_bits, tau, Z, N, dtype=F16) -> np.ndarray:
    """fp32-accumulate / round-once CPU port (OpenMP); the timed CPU baseline."""
    y = np.empty(N, dtype=np.uint16)
    _check(lib().teal_oracle_fast_sparse_gemv(_u16(x_bits), _u16(wT_bits), _u16(y), tau, Z, N, dtype), "fast")
    return y


def fast_qkv_gemv(x_bits, wT_bits, tq, tk, tv, Z, N, N_q, N_kv, dtype=F16) -> np.ndarray:
    y = np.empty(N, dtype=np.uint16)
    _check(lib().teal_oracle_fast_qkv_gemv(_u16(x_bits), _u16(wT_bits), _u16(y), tq, tk, tv, Z, N, N_q, N_kv,
                                           dtype), "fast_qkv")
    return y


def fast_dense_gemv(x_bits, wT_bits, Z, N, dtype=F16) -> np.ndarray:
    y = np.empty(N, dtype=np.uint16)
    _check(lib().teal_oracle_fast_dense_gemv(_u16(x_bits), _u16(wT_bits), _u16(y), Z, N, dtype), "dense")
    return y


class Mat:
    """Resident-matrix CPU baseline (teal_oracle_mat_*): W^T prepared once — tile-major, each region first touched by the
    thread that streams it, scratch preallocated — then `gemv(x_bits, tau)` per call (tau < 0: every row, the dense path).
    The OpenMP width must not change between creation and use."""

    def __init__(self, wT_bits, Z, N, dtype=F16):
        self.Z, self.N, self.dtype = int(Z), int(N), int(dtype)
        self._h = lib().teal_oracle_mat_create(_u16(wT_bits), self.Z, self.N, self.dtype)
        if not self._h:
            raise MemoryError("teal_oracle_mat_create failed")
        self._y = np.empty(self.N, dtype=np.uint16)

    def gemv(self, x_bits, tau) -> np.ndarray:
        _check(lib().teal_oracle_mat_gemv(self._h, _u16(x_bits), _u16(self._y), float(tau)), "mat_gemv")
        return self._y

    def close(self):
        if self._h:
            lib().teal_oracle_mat_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_read_gbs(nbytes: int = 2 << 30, reps: int = 3) -> float:
    """GB/s of a plain parallel sum over `nbytes` of first-touched memory at the current OpenMP width (the host's ceiling)."""
    return float(abs(lib().teal_oracle_host_read_gbs(int(nbytes), int(reps))))


def num_threads() -> int:
    return int(lib().teal_oracle_num_threads())


def set_threads(n: int) -> None:
    lib().teal_oracle_set_threads(int(n))


def pick_threads(x_bits, wT_bits, tau, Z, N, dtype=F16, candidates=(1, 2, 4, 8, 16, 32, 64, 128)) -> int:
    """containers often expose more logical CPUs than their quota: time one GEMV per thread count and
    keep the fastest (the CPU baseline reports the count it actually used)."""
    import os
    import time
    best, best_t = 1, float("inf")
    ncpu = os.cpu_count() or 1
    for n in candidates:
        if n > ncpu:
            break
        set_threads(n)
        fast_sparse_gemv(x_bits, wT_bits, tau, Z, N, dtype)
        t0 = time.perf_counter()
        fast_sparse_gemv(x_bits, wT_bits, tau, Z, N, dtype)
        fast_sparse_gemv(x_bits, wT_bits, tau, Z, N, dtype)
        t = (time.perf_counter() - t0) / 2
        if t < best_t:
            best, best_t = n, t
    set_threads(best)
    return best


# ----------------------------------------------------------------------------------------
# threshold math (load-time, fp32) — restated in numpy from gpt-fast/distribution.py:17-66
# ----------------------------------------------------------------------------------------
def icdf_np(counts: np.ndarray, centers: np.ndarray, q: float) -> float:
    """Distribution.icdf(q) with torch's fp32 arithmetic restated in numpy.

    total = counts.sum() (fp32), cum = cumsum(counts) (fp32), target = q*total (fp32),
    idx = searchsorted(cum, target) (left), linear interpolation between centres idx-1, idx.
    torch.cumsum / sum on CPU fp32 accumulate sequentially in fp32 for cumsum; `sum` uses
    a vectorised pairwise order, so callers pass torch-produced totals when bit-parity with
    the fixtures matters (tests/test_thresholds.py uses teal_amd.distribution for that and
    this function only as an independent cross-check within 1 ulp).
    """
    counts = np.asarray(counts, dtype=np.float32)
    centers = np.asarray(centers, dtype=np.float32)
    cum = np.cumsum(counts, dtype=np.float32)
    total = np.float32(counts.sum(dtype=np.float32))
    target = np.float32(q) * total
    idx = int(np.searchsorted(cum, target, side="left"))
    if idx == 0:
        return float(centers[0])
    if idx == len(centers):
        return float(centers[-1])
    lo_c, hi_c = cum[idx - 1], cum[idx]
    lo_v, hi_v = centers[idx - 1], centers[idx]
    frac = np.float32(target - lo_c) / np.float32(hi_c - lo_c)
    return float(np.float32(lo_v + frac * np.float32(hi_v - lo_v)))
