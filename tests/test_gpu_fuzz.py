"""GPU: seeded random sweep of the sparse GEMV boundary — odd vector lengths, ragged column counts, thresholds from
"keep everything" to "drop everything", three-threshold qkv splits, both dtypes, 16-bit and int8 weights, padded and
unpadded row strides, every compaction mode — against the oracle's double-precision truth and its index set."""
import numpy as np
import pytest
import torch

from helpers import bits_from_torch, colmajor_weight, lib_for, tolerance, torch_from_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def K():
    from teal_amd.kernels import sparse_gemv
    return sparse_gemv


def _cases(n, seed):
    r = np.random.RandomState(seed)
    out = []
    for i in range(n):
        Z = int(r.choice([64, 65, 127, 128, 200, 1000, 1024, 2048, 3000, 4096, 5120, 8192, 11008, 13824]))
        N = 8 * int(r.randint(1, 520))
        dtype = int(r.randint(0, 2))
        kind = r.choice(["one", "qkv"])
        taus = [float(r.choice([-1.0, 0.0, 0.1, 0.5, 1.0, 1.9, 50.0])) for _ in range(3)]
        kv = 0
        if kind == "qkv" and N >= 48:
            kv = 8 * int(r.randint(1, N // 24 + 1))
        out.append((i, Z, N, dtype, taus, kv, int(r.randint(0, 3)), bool(r.randint(0, 2)), bool(r.randint(0, 2))))
    return out


# (TEAL_FUZZ_CASES / TEAL_FUZZ_SEED: a longer or different sweep for a one-off run; the defaults are what the suite runs)
import os  # noqa: E402


@pytest.mark.parametrize("case", _cases(int(os.environ.get("TEAL_FUZZ_CASES", "48")), int(os.environ.get("TEAL_FUZZ_SEED", "20260927"))), ids=lambda c: f"{c[0]}-Z{c[1]}-N{c[2]}-d{c[3]}-kv{c[5]}-wl{c[6]}-{'i8' if c[7] else 'w16'}")
def test_random_gemv_cases_vs_truth(oracle, case):
    import contextlib
    i, Z, N, dtype, taus, kv, wl, int8, padded = case
    xb = oracle.hash_uniform(Z, 1000 + i, 4.0, dtype)          # U(-2, 2)
    x = torch_from_bits(xb, dtype, DEV).view(1, 1, Z)
    tq, tk, tv = taus if kv else (taus[0],) * 3
    N_q = N - 2 * kv
    to_tau = lambda t: t if t >= 0 else float("-inf")          # noqa: E731
    stack = contextlib.ExitStack()
    stack.enter_context(lib_for(wave_local=(wl if wl < 2 else 1)))  # workgroup-wide list: a switch of the diagnostics build
    try:
        if int8:
            u = oracle.from_bits(oracle.hash_uniform_c(N * Z, 2000 + i, 2.0, 0), 0).reshape(N, Z)
            q = np.clip(np.round(u * 127.0), -127, 127).astype(np.int8)
            scb = oracle.to_bits(((1 + np.arange(N) % 5) * 1e-3).astype(np.float32), dtype)
            pad = 128 if padded else 0
            buf = torch.zeros(Z, N + pad, dtype=torch.int8, device=DEV)
            buf[:, :N] = torch.from_numpy(np.ascontiguousarray(q.T)).to(DEV)
            y = K().qkv_gemv_int8(x, buf[:, :N].T, torch_from_bits(scb, dtype, DEV), to_tau(tq), to_tau(tk), to_tau(tv), 0, kv)
            truth = oracle.int8_truth64(xb, q, scb, tq, tk, tv, N_q, kv, dtype)
        else:
            wb = oracle.hash_uniform_c(Z * N, 2000 + i, 0.1, dtype)
            W = colmajor_weight(wb, Z, N, dtype, DEV)
            if padded:
                buf = torch.zeros(Z, N + 64, dtype=W.dtype, device=DEV)
                buf[:, :N] = W.T
                W = buf[:, :N].T
            y = K().qkv_gemv(x, W, to_tau(tq), to_tau(tk), to_tau(tv), 0, kv)
            truth = oracle.truth64(xb, wb, Z, N, tq, tk, tv, N_q, kv, dtype)
        got = oracle.from_bits(bits_from_torch(y.view(-1)), dtype)
        bad = np.abs(got - truth) > tolerance(oracle, truth, dtype)
        assert not bad.any(), (case, int(bad.sum()), float(np.abs(got - truth).max()))
        # the index set itself, bit-exact, for the first threshold
        if tq >= 0:
            idx, n = K().compact(x, tq)
            want = oracle.compact(xb, tq, dtype)
            assert n == len(want) and np.array_equal(idx.cpu().numpy(), want)
    finally:
        stack.close()
