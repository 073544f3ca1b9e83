"""GPU (MI355X): the HIP path, called through the C ABI, against the oracle and the golden fixtures.

Bars (BASELINE.json north_star / SURVEY §8(c)):
  * index sets: bit-exact vs {m : float32(|x[m]|) > float32(tau)}, ascending;
  * GEMV: |y - truth64| <= 1e-3*max(1,|truth64|) + 1 ulp_out(truth64)   AND
          max|y - truth64| <= max|y_ref - truth64| (never worse than the reference kernel's own
          fp16-atomic result on the same input, from the fixtures);
  * bit-reproducible run to run (no atomics).
"""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import (GOLDEN, bits_from_torch, colmajor_weight, kat_weights, load_kat, ref_keys, tolerance,
                     torch_from_bits, with_diagnostics)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

GEMV_KATS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "kat_gemv_*.npz")))
QKV_KATS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "kat_qkv_*.npz")))
INDEX_KATS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "kat_index_*.npz")))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from teal_amd import runtime
    assert runtime.init() > 0
    yield


def K():
    import teal_amd.kernels as k
    return k


def ref_final(O, y_ref_f16_bits, dtype):
    """what the reference's wrapper returns: the fp16 kernel output, cast to x.dtype when x is
    bf16 (kernels/sparse_gemv.py:138-140)."""
    v = O.from_bits(y_ref_f16_bits, 0)
    if dtype == 1:
        v = O.from_bits(O.to_bits(v, 1), 1)
    return v.astype(np.float64)


def check_gemv(O, y_bits, truth, dtype, ref_err=None, what=""):
    y = O.from_bits(y_bits, dtype).astype(np.float64)
    err = np.abs(y - truth)
    tol = tolerance(O, truth, dtype)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} columns outside tolerance, worst {err.max():.3e} (tol {tol[err.argmax()]:.3e})"
    if ref_err is not None:
        assert err.max() <= ref_err + 1e-12, f"{what}: worse than the reference kernel ({err.max():.3e} > {ref_err:.3e})"
    return err.max()


# ---------------------------------------------------------------------------------- index sets
@pytest.mark.parametrize("name", INDEX_KATS + GEMV_KATS)
def test_compact_bit_exact_golden(oracle, name):
    k = load_kat(name)
    dtype, tau = int(k["dtype"]), float(k["tau"])
    x = torch_from_bits(k["x"], dtype, DEV)
    idx, n = K().compact(x, tau)
    assert n == k["kept"].size
    assert np.array_equal(idx.cpu().numpy(), k["kept"])


@pytest.mark.parametrize("Z", [1, 63, 64, 65, 1000, 4096, 11008, 14336, 28672, 65536])
@pytest.mark.parametrize("dtype", [0, 1])
def test_compact_bit_exact_random(oracle, Z, dtype):
    rng = np.random.default_rng(Z * 2 + dtype)
    xb = oracle.to_bits(rng.standard_normal(Z).astype(np.float32), dtype)
    for tau in (0.0, 0.3, 0.6744897, 5.0, -1.0):
        idx, n = K().compact(torch_from_bits(xb, dtype, DEV), tau)
        assert np.array_equal(idx.cpu().numpy(), oracle.compact(xb, tau, dtype)), (Z, dtype, tau)


def test_compact_boundary_values(oracle):
    k = load_kat("kat_boundary.npz")
    for xb in (k["x"], k["nan_x"], k["inf_x"]):
        x = torch_from_bits(xb, 0, DEV)
        for name in ("tau_probe", "tau_zero", "tau_tiny", "tau_exact_x", "tau_below_x"):
            tau = float(k[f"{name}_tau"])
            idx, _ = K().compact(x, tau)
            assert np.array_equal(idx.cpu().numpy(), oracle.compact(xb, tau, 0))
    idx, _ = K().compact(torch_from_bits(k["x"], 0, DEV), float(k["tau_probe_tau"]))
    assert 0 in idx.cpu().numpy()  # fp16(0.1) kept at tau=0.09997: the fp32 rule, not torch's fp16 one


# ---------------------------------------------------------------------------------- GEMV vs golden
@pytest.mark.parametrize("name", GEMV_KATS)
def test_sparse_gemv_golden(oracle, name):
    k = load_kat(name)
    Z, N, dtype, tau = int(k["Z"]), int(k["N"]), int(k["dtype"]), float(k["tau"])
    wb = kat_weights(oracle, k)
    x = torch_from_bits(k["x"], dtype, DEV).view(1, 1, Z)
    W = colmajor_weight(wb, Z, N, dtype, DEV)
    truth = k["y_truth64"]
    ref_err = max(np.abs(ref_final(oracle, k[r], dtype) - truth).max() for r in ref_keys(k))
    y = K().splitk_sparse_gemv(x, W, tau, 0)
    assert y.shape == (1, 1, N) and y.dtype == x.dtype
    e = check_gemv(oracle, bits_from_torch(y.view(-1)), truth, dtype, ref_err, name)
    # deterministic: second call is bit-identical
    y2 = K().splitk_sparse_gemv(x, W, tau, 0)
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16))
    # through the registered op as monkeypatch_layer installs it
    op = K().SparseGEMV.initialize("sparse_gemv", "cuda").operator(True)
    y3 = op(x, W, tau, 0)
    assert torch.equal(y.view(torch.int16), y3.view(torch.int16))
    print(f"{name}: max|err| {e:.2e} (reference kernel {ref_err:.2e})")


@pytest.mark.parametrize("name", QKV_KATS)
def test_qkv_gemv_golden(oracle, name):
    k = load_kat(name)
    Z, N, N_q, N_kv, dtype = (int(k[n]) for n in ("Z", "N", "N_q", "N_kv", "dtype"))
    tq, tk, tv = float(k["tau_q"]), float(k["tau_k"]), float(k["tau_v"])
    wb = kat_weights(oracle, k)
    x = torch_from_bits(k["x"], dtype, DEV).view(1, 1, Z)
    W = colmajor_weight(wb, Z, N, dtype, DEV)
    truth = k["y_truth64"]
    ref_err = max(np.abs(ref_final(oracle, k[r], dtype) - truth).max() for r in ref_keys(k))
    y = K().qkv_gemv(x, W, tq, tk, tv, 0, N_kv)
    check_gemv(oracle, bits_from_torch(y.view(-1)), truth, dtype, ref_err, name)
    op = K().SparseQKVGEMV.initialize("sparse_qkv_gemv", "cuda").operator(True)
    assert torch.equal(op(x, W, tq, tk, tv, 0, N_kv).view(torch.int16), y.view(torch.int16))


def test_boundary_nan_inf_propagate_like_reference(oracle):
    k = load_kat("kat_boundary.npz")
    Z = k["x"].size
    eye = np.eye(Z, dtype=np.float16).view(np.uint16).reshape(-1).copy()
    W = colmajor_weight(eye, Z, Z, 0, DEV)
    for name in ("tau_probe", "tau_zero", "tau_tiny", "tau_exact_x", "tau_below_x"):
        y = K().splitk_sparse_gemv(torch_from_bits(k["x"], 0, DEV).view(1, 1, Z), W, float(k[f"{name}_tau"]), 0)
        # identity weights: no rounding involved, so the reference output is reproduced bit-for-bit
        assert np.array_equal(bits_from_torch(y.view(-1)), k[f"{name}_y"]), name
    yn = K().splitk_sparse_gemv(torch_from_bits(k["nan_x"], 0, DEV).view(1, 1, Z), W, float(k["tau_probe_tau"]), 0)
    assert torch.isnan(yn).all()  # the reference's 0*NaN poisons every column
    yi = K().splitk_sparse_gemv(torch_from_bits(k["inf_x"], 0, DEV).view(1, 1, Z), W, float(k["tau_probe_tau"]), 0).view(-1)
    ri = oracle.from_bits(k["inf_y"], 0)
    assert np.array_equal(np.isnan(ri), torch.isnan(yi).cpu().numpy()) and torch.isinf(yi[7])


# ---------------------------------------------------------------------------------- shapes / geometry
@pytest.mark.parametrize("Z,N", [(64, 8), (100, 72), (257, 520), (1000, 1000), (4096, 4104), (3000, 11008)])
@pytest.mark.parametrize("dtype", [0, 1])
def test_ragged_shapes_vs_oracle(oracle, Z, N, dtype):
    xb = oracle.hash_uniform(Z, 11 + Z, 4.0, dtype)
    wb = oracle.hash_uniform_c(Z * N, 13 + N, 0.1, dtype)
    x = torch_from_bits(xb, dtype, DEV).view(1, 1, Z)
    W = colmajor_weight(wb, Z, N, dtype, DEV)
    for tau in (-1.0, 0.0, 0.9, 1.7, 100.0):  # dense ... everything dropped
        truth = oracle.truth64(xb, wb, Z, N, tau, dtype=dtype)
        y = K().splitk_sparse_gemv(x, W, tau, 0)
        check_gemv(oracle, bits_from_torch(y.view(-1)), truth, dtype, None, f"{Z}x{N} tau={tau}")
    yd = K().dense_gemv(x, W)
    check_gemv(oracle, bits_from_torch(yd.view(-1)), oracle.truth64(xb, wb, Z, N, -1.0, dtype=dtype), dtype)


@pytest.mark.parametrize("wave_local", [1, 0])
@with_diagnostics
def test_every_launch_geometry_agrees(oracle, wave_local):
    """all (lanes_per_row, split) geometries compute the same GEMV (within rounding), with the wave-local compaction
    and with the workgroup-wide even-share list, through the lean and the general kernel (incl. the single-launch
    split-K with arrival tickets)."""
    from teal_amd import _lib
    L = _lib.load()
    L.teal_set_wave_local(wave_local)
    Z, N, dtype = 1536, 1280, 0
    xb = oracle.hash_uniform(Z, 5, 4.0, dtype)
    wb = oracle.hash_uniform_c(Z * N, 6, 0.1, dtype)
    x = torch_from_bits(xb, dtype, DEV).view(1, 1, Z)
    W = colmajor_weight(wb, Z, N, dtype, DEV)
    truth = oracle.truth64(xb, wb, Z, N, 1.0, 0.5, 1.5, N - 2 * 256, 256, dtype)  # qkv_gemv: N_q = N - 2*kv_size
    try:
        assert L.teal_set_tuning(8, 8, 1, 4) != 0 and L.teal_set_tuning(8, 16, 1, 8) != 0  # sweep-only variants are gone
        for fast in (1, 0):  # lean kernel where the shape qualifies (8 / 16 lanes, <= 8 slices), general kernel otherwise
            L.teal_set_fast(fast)
            for lpr in (8, 16, 32, 64):
                for split in (1, 2, 3, 5, 8, 32):
                    assert L.teal_set_tuning(lpr, 16, split, 4) == 0
                    y = K().qkv_gemv(x, W, 1.0, 0.5, 1.5, 0, 256)
                    check_gemv(oracle, bits_from_torch(y.view(-1)), truth, dtype, None, f"cfg {lpr},{split} fast={fast}")
    finally:
        L.teal_set_tuning(0, 0, 0, 0)
        L.teal_set_wave_local(1)
        L.teal_set_fast(1)


@pytest.mark.parametrize("Z,N,dtype,s", [(8192, 8192, 0, 0.5), (8192, 28672, 0, 0.5), (28672, 8192, 0, 0.5),
                                         (14336, 4096, 1, 0.4), (4096, 12288, 0, 0.5)])
def test_full_size_shapes_vs_cpu_port(oracle, Z, N, dtype, s):
    """BASELINE full sizes (70B / 8B shapes): HIP vs the oracle's fp32 CPU port + truth64 on a column sample."""
    xb = oracle.hash_uniform(Z, 21 + Z, 4.0, dtype)
    wb = oracle.hash_uniform_c(Z * N, 23 + N, 0.08, dtype)
    tau = 2.0 * s  # x ~ U(-2, 2): P(|x| <= tau) = s
    x = torch_from_bits(xb, dtype, DEV).view(1, 1, Z)
    W = colmajor_weight(wb, Z, N, dtype, DEV)
    y = bits_from_torch(K().splitk_sparse_gemv(x, W, tau, 0).view(-1))
    truth = oracle.truth64(xb, wb, Z, N, tau, dtype=dtype)
    check_gemv(oracle, y, truth, dtype, None, f"{Z}x{N}")
    cpu = oracle.from_bits(oracle.fast_sparse_gemv(xb, wb, tau, Z, N, dtype), dtype)
    # both round an fp32 sum once, in different summation orders
    assert (np.abs(oracle.from_bits(y, dtype) - cpu) <= 2 * tolerance(oracle, truth, dtype)).all()


def test_linearity_property_full_size(oracle):
    """size-independent property: gemv(a*x) == a*gemv(x) exactly for a power of two (kept set fixed by scaling tau)."""
    Z, N = 4096, 11008
    xb = oracle.hash_uniform(Z, 31, 2.0, 0)
    W = colmajor_weight(oracle.hash_uniform_c(Z * N, 32, 0.05, 0), Z, N, 0, DEV)
    x = torch_from_bits(xb, 0, DEV).view(1, 1, Z)
    y1 = K().splitk_sparse_gemv(x, W, 0.5, 0)
    y2 = K().splitk_sparse_gemv(x * 2, W, 1.0, 0)
    assert torch.equal((y1 * 2).view(torch.int16), y2.view(torch.int16))
    # dropping rows by zeroing them == raising the threshold
    xz = torch.where(x.abs().float() > 0.5, x, torch.zeros_like(x))
    y3 = K().splitk_sparse_gemv(xz, W, 0.0, 0)
    assert torch.equal(y1.view(torch.int16), y3.view(torch.int16))


# ---------------------------------------------------------------------------------- fusion, graphs, errors
def test_gateup_silu_fusion_equals_unfused_sequence(oracle):
    for dtype, Z, N in ((0, 4096, 11008), (1, 4096, 14336), (0, 512, 1032)):
        xb = oracle.hash_uniform(Z, 41, 4.0, dtype)
        x = torch_from_bits(xb, dtype, DEV).view(1, 1, Z)
        W1 = colmajor_weight(oracle.hash_uniform_c(Z * N, 42, 0.08, dtype), Z, N, dtype, DEV)
        W3 = colmajor_weight(oracle.hash_uniform_c(Z * N, 43, 0.08, dtype), Z, N, dtype, DEV)
        g = K().splitk_sparse_gemv(x, W1, 0.9, 0)
        u = K().splitk_sparse_gemv(x, W3, 1.1, 0)
        want = torch.nn.functional.silu(g) * u  # gpt-fast/model.py:258-259
        got = K().sparse_gateup_silu(x, W1, W3, 0.9, 1.1)
        diff = (got.float() - want.float()).abs()
        ulp = torch.from_numpy(oracle.ulp16(want.float().cpu().numpy(), dtype)).to(DEV)
        # silu's expf may round differently from torch's by 1 ulp of the fp16 activation
        # + a 1-ulp difference of gate (different fp32 summation order) scaled by |up|
        assert (diff.view(-1) <= 2 * ulp.view(-1).float() + 1e-3).all(), float(diff.max())
        assert (diff == 0).float().mean() > 0.9


def test_hipgraph_capture_and_replay(oracle):
    Z, N = 4096, 4096
    xb = oracle.hash_uniform(Z, 51, 4.0, 0)
    W = colmajor_weight(oracle.hash_uniform_c(Z * N, 52, 0.08, 0), Z, N, 0, DEV)
    x = torch_from_bits(xb, 0, DEV).view(1, 1, Z)
    eager = K().splitk_sparse_gemv(x, W, 1.0, 0).clone()
    xs = x.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        K().splitk_sparse_gemv(xs, W, 1.0, 0)  # warm-up on the side stream
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = torch.ops.teal.sparse_gemv(xs, W, 1.0, 0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), eager.view(torch.int16))
    xs.copy_(x * 0.5)  # new activations, same graph: the kept set changes on-device
    g.replay()
    torch.cuda.synchronize()
    want = K().splitk_sparse_gemv(x * 0.5, W, 1.0, 0)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("Z,N,dtype", [(4096, 4096, 0), (11008, 4096, 0), (4096, 11008, 1), (1000, 1000, 0), (8192, 8192, 0)])
@with_diagnostics
def test_wave_local_and_list_paths_agree(oracle, Z, N, dtype):
    """the two compaction strategies keep the same rows; results agree to fp32 summation order"""
    from teal_amd import _lib
    L = _lib.load()
    xb = oracle.hash_uniform(Z, 71, 4.0, dtype)
    wb = oracle.hash_uniform_c(Z * N, 72, 0.08, dtype)
    x = torch_from_bits(xb, dtype, DEV).view(1, 1, Z)
    W = colmajor_weight(wb, Z, N, dtype, DEV)
    truth = oracle.truth64(xb, wb, Z, N, 1.0, dtype=dtype)
    try:
        outs = []
        for wl in (1, 0):
            L.teal_set_wave_local(wl)
            y = K().splitk_sparse_gemv(x, W, 1.0, 0)
            check_gemv(oracle, bits_from_torch(y.view(-1)), truth, dtype, None, f"wl={wl}")
            outs.append(y.float())
        assert (outs[0] - outs[1]).abs().max() <= 2 * float(oracle.ulp16(np.abs(truth).max(), dtype))
    finally:
        L.teal_set_wave_local(1)


def test_padded_row_stride_equals_unpadded(oracle):
    """weights with strides (1, N + 64) (what monkeypatch_layer installs) give the same bits as (1, N)."""
    Z, N = 4096, 4096
    xb = oracle.hash_uniform(Z, 61, 4.0, 0)
    wb = oracle.hash_uniform_c(Z * N, 62, 0.08, 0)
    x = torch_from_bits(xb, 0, DEV).view(1, 1, Z)
    W = colmajor_weight(wb, Z, N, 0, DEV)
    buf = torch.zeros(Z, N + 64, device=DEV, dtype=torch.float16)
    buf[:, :N] = W.T
    Wp = buf[:, :N].T
    assert Wp.stride() == (1, N + 64)
    for tau in (1.0, -1.0):
        a = K().splitk_sparse_gemv(x, W, tau, 0)
        b = K().splitk_sparse_gemv(x, Wp, tau, 0)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    q1 = K().qkv_gemv(x, W, 0.5, 1.0, 1.5, 0, 1024)
    q2 = K().qkv_gemv(x, Wp, 0.5, 1.0, 1.5, 0, 1024)
    assert torch.equal(q1.view(torch.int16), q2.view(torch.int16))
    assert torch.equal(K().dense_gemv(x, W).view(torch.int16), K().dense_gemv(x, Wp).view(torch.int16))


def test_prefill_falls_back_to_dense_matmul():
    Z, N = 256, 512
    x = torch.randn(1, 5, Z, device=DEV, dtype=torch.float16)
    W = (torch.randn(N, Z, device=DEV, dtype=torch.float16) * 0.05).T.contiguous().T
    y = torch.ops.teal.sparse_gemv(x, W, 0.5, 0)  # kernels/sparse_gemv.py:271: seq_len > 1 -> matmul, no masking
    assert torch.allclose(y, torch.matmul(x, W.T))


@with_diagnostics  # (a forced split-K geometry provokes the workspace error)
def test_c_abi_error_codes():
    from teal_amd import _lib, runtime
    L = _lib.load()
    x = torch.zeros(64, device=DEV, dtype=torch.float16)
    w = torch.zeros(64 * 64, device=DEV, dtype=torch.float16)
    y = torch.zeros(64, device=DEV, dtype=torch.float16)
    ws = runtime.reserve_workspace(64, 64)
    st = runtime.stream_ptr()
    ok = L.teal_sparse_gemv(x.data_ptr(), w.data_ptr(), y.data_ptr(), 0.1, 64, 64, 0, ws.data_ptr(), ws.numel() * 4, st)
    assert ok == 0
    assert L.teal_sparse_gemv(None, w.data_ptr(), y.data_ptr(), 0.1, 64, 64, 0, ws.data_ptr(), ws.numel() * 4, st) == -1
    assert L.teal_sparse_gemv(x.data_ptr(), w.data_ptr(), y.data_ptr(), 0.1, 64, 64, 7, ws.data_ptr(), ws.numel() * 4, st) == -2
    assert L.teal_sparse_gemv(x.data_ptr(), w.data_ptr(), y.data_ptr(), 0.1, 64, 60, 0, ws.data_ptr(), ws.numel() * 4, st) == -3
    assert L.teal_sparse_gemv(x.data_ptr(), w.data_ptr() + 2, y.data_ptr(), 0.1, 64, 64, 0, ws.data_ptr(), ws.numel() * 4, st) == -4
    assert L.teal_set_tuning(8, 16, 4, 4) == 0
    try:
        assert L.teal_sparse_gemv(x.data_ptr(), w.data_ptr(), y.data_ptr(), 0.1, 64, 64, 0, ws.data_ptr(), 16, st) == -5
    finally:
        L.teal_set_tuning(0, 0, 0, 0)
    with pytest.raises(TypeError):
        import teal_amd.kernels as k
        k.splitk_sparse_gemv(torch.zeros(1, 1, 64, device=DEV), torch.zeros(64, 64, device=DEV).T.contiguous().T, 0.1, 0)
    torch.cuda.synchronize()


def test_deja_vu_comparator_computes_the_same_masked_gemv():
    """the benchmark's Deja Vu comparator (teal_cmp_flag_gemv: precomputed flags, fp32 atomics into a zeroed output;
    scripts/benchmark_gemv.py:32-107 of the reference) against the product kernel on the benchmark's input law"""
    from teal_amd import _lib, runtime
    L = _lib.load()
    runtime.init()
    for Z, N, ld, tau in ((4096, 4096, 4160, 0.25), (1000, 520, 520, 0.1), (4096, 14336, 14400, -1.0)):
        g = torch.Generator(device=DEV).manual_seed(Z + N)
        x = (torch.rand(Z, device=DEV, generator=g) - 0.5).half()
        buf = torch.zeros(Z, ld, device=DEV, dtype=torch.float16)
        buf[:, :N] = (torch.rand(Z, N, device=DEV, generator=g) - 0.5).half()
        y32 = torch.full((N,), 7.0, device=DEV, dtype=torch.float32)
        flags = torch.zeros(Z, device=DEV, dtype=torch.uint8)
        assert L.teal_cmp_flag_gemv(x.data_ptr(), buf.data_ptr(), ld, y32.data_ptr(), flags.data_ptr(), tau, Z, N, 0, runtime.stream_ptr()) == 0
        keep = x.float().abs() > tau
        assert torch.equal(flags.bool(), keep)
        want = (buf[:, :N].double() * (x.double() * keep)[:, None]).sum(0)
        assert torch.allclose(y32.double(), want, atol=1e-3, rtol=1e-4), float((y32.double() - want).abs().max())


@with_diagnostics  # (the fall-back of a RoPE request needs the general kernel forced)
def test_c_abi_round4_additions_qkv_rope_and_act_seg0():
    """TEAL_OUT_QKV_ROPE straight through the C ABI: the epilogue's rotated q / appended k, v against a torch restatement of
    gpt-fast/model.py:170-178 applied to the SAME launch's slab-mode projection (general kernel: the request falls back to
    slabs and says so); argument rules of act_seg0 / gate_activated."""
    import ctypes
    from teal_amd import _lib, runtime
    from teal_amd.gpt_fast.engine import GemvIn, GemvOut, TEAL_IN_PLAIN, TEAL_IN_RESID_NORM, TEAL_OUT_QKV_ROPE, TEAL_OUT_ROUNDED, TEAL_OUT_SLABS
    L = _lib.load()
    runtime.init()
    dim, nkv, hd, max_seq, P = 4096, 32, 128, 64, 9
    kv = nkv * hd
    nqkv, ld = dim + 2 * kv, dim + 2 * kv + 64
    g = torch.Generator(device=DEV).manual_seed(12)
    for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
        w = ((torch.rand(dim, ld, device=DEV, generator=g) - 0.5) * 0.05).to(dt)
        resid = torch.randn(dim, device=DEV, generator=g).to(dt)
        normw = (1.0 + 0.1 * torch.randn(dim, device=DEV, generator=g)).to(dt)
        ang = torch.outer(torch.arange(max_seq, device=DEV).float(), 1.0 / (10000 ** (torch.arange(0, hd, 2, device=DEV).float() / hd)))
        rope = torch.stack((torch.cos(ang), torch.sin(ang)), dim=-1).to(dt).contiguous()
        pos = torch.tensor([P], device=DEV, dtype=torch.int32)
        kc = torch.zeros(nkv, max_seq, hd, device=DEV, dtype=dt)
        vc = torch.zeros_like(kc)
        q = torch.zeros(nqkv, device=DEV, dtype=dt)
        slabs = torch.zeros(8 * nqkv, device=DEV, dtype=torch.float32)
        hout = torch.zeros(dim, device=DEV, dtype=dt)
        ws = runtime.new_workspace(dim, nqkv)
        gin = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=resid.data_ptr(), slabs=None, nslabs=0, slabs_interleaved=1,
                     norm_weight=normw.data_ptr(), eps=1e-5, resid_out=hout.data_ptr())
        tau = 0.6

        def out(mode):
            o = GemvOut()
            o.nseg, o.mode = 3, mode
            for i, (c0, nc) in enumerate(((0, dim), (dim, kv), (dim + kv, kv))):
                o.w[i], o.ld[i], o.col0[i], o.ncols[i], o.tau[i], o.y[i] = w.data_ptr(), ld, c0, nc, tau, q.data_ptr() + 2 * c0
            o.slabs, o.slabs_bytes, o.slabs_interleaved = slabs.data_ptr(), slabs.numel() * 4, 1
            o.rope, o.rope_pos, o.k_cache, o.v_cache = rope.data_ptr(), pos.data_ptr(), kc.data_ptr(), vc.data_ptr()
            o.rope_head_dim, o.rope_max_seq = hd, max_seq
            return o

        n = ctypes.c_int(-1)
        args = (ctypes.byref(gin), None, dim, code, ws.data_ptr(), ws.numel() * 4, ctypes.byref(n), runtime.stream_ptr())
        o = out(TEAL_OUT_QKV_ROPE)
        assert L.teal_fused_gemv(args[0], ctypes.byref(o), *args[2:]) == 0 and n.value == 0  # the epilogue ran
        got_q, got_k, got_v = q[:dim].clone(), kc[:, P].clone(), vc[:, P].clone()
        assert float(kc[:, :P].abs().max()) == 0 and float(kc[:, P + 1:].abs().max()) == 0  # only the token's row was written
        o = out(TEAL_OUT_SLABS)
        assert L.teal_fused_gemv(args[0], ctypes.byref(o), *args[2:]) == 0 and n.value == 1
        proj = slabs[: nqkv * 4].view(nqkv, 4)[:, 0].to(dt)  # one slab, rounded like the epilogue rounds it
        c_, s_ = rope[P, :, 0].float(), rope[P, :, 1].float()

        def rot(v):  # interleaved pairs, fp32 with the helper's fused multiply-add (model.py: apply_rotary_emb)
            v = v.float().view(-1, hd // 2, 2)
            a = torch.addcmul(-(v[..., 1] * s_), v[..., 0], c_)
            b = torch.addcmul(v[..., 0] * s_, v[..., 1], c_)
            return torch.stack((a, b), dim=-1).reshape(-1)

        want_q, want_k = rot(proj[:dim]), rot(proj[dim:dim + kv])
        ulp = lambda t: t.abs().clamp_min(6e-5) * (2.0 ** -10 if dt == torch.float16 else 2.0 ** -7)  # noqa: E731
        assert bool(((got_q.float() - want_q).abs() <= ulp(want_q)).all()), "rotated q"
        assert bool(((got_k.float().view(-1) - want_k).abs() <= ulp(want_k)).all()), "rotated k row"
        assert torch.equal(got_v.view(-1).view(torch.int16), proj[dim + kv:].view(torch.int16)), "v row: the rounded projection, bit for bit"
        # the general kernel has no such epilogue: the request falls back to the slabs and reports it
        L.teal_set_fast(0)
        try:
            o = out(TEAL_OUT_QKV_ROPE)
            assert L.teal_fused_gemv(args[0], ctypes.byref(o), *args[2:]) == 0 and n.value >= 1
            o.slabs = None
            assert L.teal_fused_gemv(args[0], ctypes.byref(o), *args[2:]) == -1  # TEAL_ERR_ARG: nowhere to fall back to
        finally:
            L.teal_set_fast(1)
        # act_seg0 belongs to rounded outputs; gate_activated / act_seg0 are refused by the int4 kernel
        o = out(TEAL_OUT_SLABS)
        o.act_seg0 = 1
        assert L.teal_fused_gemv(args[0], ctypes.byref(o), *args[2:]) == -1
        o = out(TEAL_OUT_ROUNDED)
        o.weight_bits, o.groupsize, o.act_seg0 = 4, 32, 1
        for i in range(3):
            o.scale[i], o.scale_ld[i] = w.data_ptr(), nqkv
        gp = GemvIn(mode=TEAL_IN_PLAIN, x=resid.data_ptr())
        assert L.teal_fused_gemv(ctypes.byref(gp), ctypes.byref(o), *args[2:]) == -1
    torch.cuda.synchronize()
