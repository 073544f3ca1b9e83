"""CPU: pin the oracle against fixtures produced by the REFERENCE itself (oracle/gen_golden.py ran
the reference's Triton kernels under TRITON_INTERPRET=1 in the build container)."""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN, kat_weights, load_kat, ref_keys

GEMV_KATS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "kat_gemv_*.npz")))
QKV_KATS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "kat_qkv_*.npz")))
INDEX_KATS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "kat_index_*.npz")))


def test_fixture_inventory():
    assert len(GEMV_KATS) >= 6 and len(QKV_KATS) >= 3 and len(INDEX_KATS) >= 3


def test_half_conversions_match_numpy(oracle):
    L = oracle.lib()
    allh = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    ok = ~np.isnan(f)
    mine = np.array([L.teal_oracle_half_to_float(int(h)) for h in allh[::7]], dtype=np.float32)
    assert np.array_equal(mine[ok[::7]], f[::7][ok[::7]])
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1, 300, 7e4)])
    vals = np.concatenate([vals, np.array([0.0, -0.0, 65504, 65520, 65519.99, 2.0**-24, 2.0**-25, 1.5 * 2.0**-25, np.inf, -np.inf], dtype=np.float32)])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    got = np.array([L.teal_oracle_float_to_half(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(got, want)
    # bf16 round-to-nearest-even
    u = vals.view(np.uint32)
    want_b = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    got_b = np.array([L.teal_oracle_float_to_bf16(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(got_b, want_b)


def test_hash_generator_c_equals_numpy(oracle):
    for dtype in (0, 1):
        a = oracle.hash_uniform(10007, seed=42, scale=0.08, dtype=dtype)
        b = oracle.hash_uniform_c(10007, seed=42, scale=0.08, dtype=dtype)
        assert np.array_equal(a, b)
    v = oracle.from_bits(oracle.hash_uniform(1 << 16, 3), 0)
    assert -0.5 <= v.min() and v.max() < 0.5 and abs(v.mean()) < 0.01


@pytest.mark.parametrize("name", GEMV_KATS)
def test_gemv_restatement_bit_exact_vs_reference(oracle, name):
    k = load_kat(name)
    Z, N, dtype, tau = int(k["Z"]), int(k["N"]), int(k["dtype"]), float(k["tau"])
    wb = kat_weights(oracle, k)
    assert np.array_equal(oracle.compact(k["x"], tau, dtype), k["kept"])
    assert np.array_equal(oracle.compact_np(k["x"], tau, dtype), k["kept"])
    keys = ref_keys(k)
    assert keys
    for key in keys:
        bm, bn = (int(v) for v in key[len("y_ref_"):].split("x"))
        mine = oracle.ref_sparse_gemv(k["x"], wb, tau, Z, N, dtype, bm, bn)
        assert np.array_equal(mine, k[key]), f"{name}:{key} restatement differs from the reference kernel"
    truth = oracle.truth64(k["x"], wb, Z, N, tau, dtype=dtype)
    assert np.array_equal(truth, k["y_truth64"])
    if Z * N <= 1 << 18:
        assert np.allclose(truth, oracle.truth64_np(k["x"], wb, Z, N, tau, dtype), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", QKV_KATS)
def test_qkv_restatement_bit_exact_vs_reference(oracle, name):
    k = load_kat(name)
    Z, N, N_q, N_kv, dtype = (int(k[n]) for n in ("Z", "N", "N_q", "N_kv", "dtype"))
    tq, tk, tv = float(k["tau_q"]), float(k["tau_k"]), float(k["tau_v"])
    wb = kat_weights(oracle, k)
    for key in ref_keys(k):
        bm, bn = (int(v) for v in key[len("y_ref_"):].split("x"))
        mine = oracle.ref_qkv_gemv(k["x"], wb, tq, tk, tv, Z, N, N_q, N_kv, dtype, bm, bn)
        assert np.array_equal(mine, k[key])
    assert np.array_equal(oracle.truth64(k["x"], wb, Z, N, tq, tk, tv, N_q, N_kv, dtype), k["y_truth64"])
    for t, kk in ((tq, "kept_q"), (tk, "kept_k"), (tv, "kept_v")):
        assert np.array_equal(oracle.compact(k["x"], t, dtype), k[kk])


@pytest.mark.parametrize("name", INDEX_KATS)
def test_index_kats(oracle, name):
    k = load_kat(name)
    assert np.array_equal(oracle.compact(k["x"], float(k["tau"]), int(k["dtype"])), k["kept"])


def test_boundary_values_decided_by_reference_kernel(oracle):
    k = load_kat("kat_boundary.npz")
    x = k["x"]
    Z = x.size
    eye = np.eye(Z, dtype=np.float16).view(np.uint16).reshape(-1).copy()
    for name in ("tau_probe", "tau_zero", "tau_tiny", "tau_exact_x", "tau_below_x"):
        tau = float(k[f"{name}_tau"])
        # the restatement reproduces the reference kernel's output bit-for-bit ...
        assert np.array_equal(oracle.ref_sparse_gemv(x, eye, tau, Z, Z, 0, 16, 16), k[f"{name}_y"])
        # ... and the kept set (minus exact zeros, which cannot be observed through y) is the rule's
        kept = oracle.compact(x, tau, 0)
        nz = kept[oracle.from_bits(x[kept], 0) != 0]
        assert np.array_equal(nz, k[f"{name}_kept_nonzero"])
    # survey probe: x = fp16(0.1) is KEPT by the kernel at tau = 0.09997 ...
    assert 0 in k["tau_probe_kept_nonzero"]
    # ... but DROPPED by SparsifyFn.apply, which compares in fp16 (utils/utils.py:51-52)
    sf = k["sparsifyfn_out"]
    assert sf[0] == 0 and np.array_equal(oracle.sparsify_fn_apply(x, float(k["sparsifyfn_tau"]), 0), sf)
    # strict '>' : tau == x drops, tau just below x (still == x in fp32) drops too
    assert 0 not in k["tau_exact_x_kept_nonzero"] and 0 not in k["tau_below_x_kept_nonzero"]
    # NaN in x poisons every output column of the reference (0 * NaN on masked rows)
    yn = oracle.ref_sparse_gemv(k["nan_x"], eye, float(k["tau_probe_tau"]), Z, Z, 0, 16, 16)
    assert np.isnan(oracle.from_bits(k["nan_y"], 0)).all() and np.isnan(oracle.from_bits(yn, 0)).all()
    # +inf is kept: its own column is inf, every other column inf * 0 = NaN
    yi = oracle.from_bits(k["inf_y"], 0)
    assert np.isinf(yi[7]) and np.isnan(np.delete(yi, 7)).all()
    assert np.array_equal(oracle.ref_sparse_gemv(k["inf_x"], eye, float(k["tau_probe_tau"]), Z, Z, 0, 16, 16), k["inf_y"])


def test_fast_port_within_tolerance_of_truth(oracle):
    from helpers import tolerance
    for name in ("kat_gemv_small_f16.npz", "kat_gemv_small_bf16.npz", "kat_gemv_wo_7b_f16.npz"):
        k = load_kat(name)
        Z, N, dtype, tau = int(k["Z"]), int(k["N"]), int(k["dtype"]), float(k["tau"])
        wb = kat_weights(oracle, k)
        y = oracle.from_bits(oracle.fast_sparse_gemv(k["x"], wb, tau, Z, N, dtype), dtype).astype(np.float64)
        assert (np.abs(y - k["y_truth64"]) <= tolerance(oracle, k["y_truth64"], dtype)).all()
    k = load_kat("kat_qkv_small_f16.npz")
    Z, N, N_q, N_kv, dtype = (int(k[n]) for n in ("Z", "N", "N_q", "N_kv", "dtype"))
    y = oracle.fast_qkv_gemv(k["x"], k["wT"], float(k["tau_q"]), float(k["tau_k"]), float(k["tau_v"]), Z, N, N_q, N_kv, dtype)
    assert (np.abs(oracle.from_bits(y, dtype) - k["y_truth64"]) <= tolerance(oracle, k["y_truth64"], dtype)).all()


@pytest.mark.parametrize("tag,dtype", [("f16", 0), ("bf16", 1)])
def test_int8_quantiser_and_module_match_reference(oracle, golden_dir, tag, dtype):
    """F8: the numpy restatement of the reference int8 quantiser is bit-exact (q, fp32 scales, scales in the model
    dtype), and its forward restatement matches the reference module's output on the masked activation."""
    k = np.load(os.path.join(golden_dir, "kat_int8.npz"))
    w = oracle.from_bits(k[f"{tag}_w"], dtype)
    q, sc = oracle.quantize_per_channel_np(w)
    assert np.array_equal(q, k[f"{tag}_q"])
    assert np.array_equal(sc.view(np.uint32), k[f"{tag}_scales_f32"].view(np.uint32))
    assert np.array_equal(oracle.to_bits(sc, dtype), k[f"{tag}_scales"])
    assert sc[3] == np.finfo(np.float32).eps and not q[3].any()           # all-zero row: eps scale, zero codes
    assert q[5].min() == -127 or q[5].min() == -128                        # negative extreme maps to the low end
    for key, tau in (("y_masked", float(k[f"{tag}_tau"])), ("y_dense", -1.0)):
        got = oracle.from_bits(oracle.int8_ref_forward(k[f"{tag}_x"], q, k[f"{tag}_scales"], tau, dtype), dtype)
        want = oracle.from_bits(k[f"{tag}_{key}"], dtype)
        # the reference sums in fp32 in an unspecified order; the restatement sums in fp64: <= 1 ulp apart
        assert np.all(np.abs(got - want) <= oracle.ulp16(np.maximum(np.abs(want), 1e-30), dtype)), key
        assert np.mean(got == want) > 0.9, key
        truth = oracle.int8_truth64(k[f"{tag}_x"], q, k[f"{tag}_scales"], tau, dtype=dtype)
        assert np.all(np.abs(want - truth) <= 1e-3 * np.maximum(1, np.abs(truth)) + 2 * oracle.ulp16(truth, dtype))


@pytest.mark.parametrize("dtype", [0, 1])
def test_resident_cpu_baseline_matches_truth(oracle, dtype):
    """the timed CPU baseline's resident-matrix form (teal_oracle_mat_*: tile-major copy, first-touch placement, no
    per-call allocation) computes the same masked GEMV as the double-precision truth, sparse and dense, ragged widths too"""
    O = oracle
    for Z, N, tau in ((1024, 1536, 0.25), (700, 1000, 0.1), (4096, 4096, -1.0)):
        x = O.hash_uniform(Z, 31 + Z, 1.0, dtype)
        W = O.hash_uniform(Z * N, 17 + N, 1.0, dtype)
        m = O.Mat(W, Z, N, dtype)
        for t in (tau, -1.0):
            got = O.from_bits(m.gemv(x, t).copy(), dtype)
            truth = O.truth64(x, W, Z, N, t, dtype=dtype)
            ulp = O.ulp16(np.maximum(np.abs(truth), 1e-30), dtype)
            assert np.all(np.abs(got - truth) <= 1e-3 * np.maximum(1.0, np.abs(truth)) + ulp), (Z, N, t)
        again = m.gemv(x, tau).copy()
        assert np.array_equal(again, m.gemv(x, tau))  # deterministic: static schedule, partials summed in block order
        m.close()
