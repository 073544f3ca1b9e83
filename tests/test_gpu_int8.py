"""GPU: int8 weight-only sparse GEMV against the oracle's double-precision truth, the
reference module's output captured in tests/golden/kat_int8.npz, and — for the fused decode engine — against the
unfused int8 module path."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, bits_from_torch, tolerance, torch_from_bits, with_diagnostics

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def K():
    from teal_amd.kernels import sparse_gemv
    return sparse_gemv


def colmajor_int8(q: np.ndarray, pad: int):
    """torch int8 [N, Z] with strides (1, N + pad) whose memory image is q^T [Z][N + pad]."""
    N, Z = q.shape
    buf = torch.zeros(Z, N + pad, dtype=torch.int8, device=DEV)
    buf[:, :N] = torch.from_numpy(np.ascontiguousarray(q.T)).to(DEV)
    return buf[:, :N].T


def hash_int8(O, N, Z, seed):
    """seeded int8 codes in [-127, 127] from the oracle's portable generator."""
    u = O.from_bits(O.hash_uniform_c(N * Z, seed, 2.0, 0), 0).reshape(N, Z)  # U(-1, 1)
    return np.clip(np.round(u * 127.0), -127, 127).astype(np.int8)


@pytest.mark.parametrize("tag,dtype", [("f16", 0), ("bf16", 1)])
def test_fixture_reference_module_outputs(oracle, tag, dtype):
    k = np.load(os.path.join(GOLDEN, "kat_int8.npz"))
    q, scb, xb, tau = k[f"{tag}_q"], k[f"{tag}_scales"], k[f"{tag}_x"], float(k[f"{tag}_tau"])
    N, Z = q.shape
    x = torch_from_bits(xb, dtype, DEV).view(1, 1, Z)
    sc = torch_from_bits(scb, dtype, DEV)
    for pad in (0, 128):
        W = colmajor_int8(q, pad)
        for key, t in (("y_masked", tau), ("y_dense", float("-inf"))):
            y = K().splitk_sparse_gemv_int8(x, W, sc, t)
            got = oracle.from_bits(bits_from_torch(y.view(-1)), dtype)
            truth = oracle.int8_truth64(xb, q, scb, t if t > -1e30 else -1.0, dtype=dtype)
            assert np.all(np.abs(got - truth) <= tolerance(oracle, truth, dtype)), (key, pad)
            want = oracle.from_bits(k[f"{tag}_{key}"], dtype)  # the reference rounds twice, the kernel once
            assert np.all(np.abs(got - want) <= 2 * oracle.ulp16(np.maximum(np.abs(want), 1e-30), dtype)), (key, pad)
            # never worse than the reference module
            assert np.abs(got - truth).max() <= np.abs(want - truth).max() + 1e-12


@pytest.mark.parametrize("Z,N,dtype,tau", [(4096, 4096, 0, 0.5), (4096, 11008, 0, 0.5), (11008, 4096, 0, 0.35),
                                           (4096, 14336, 1, 0.4), (8192, 1024, 0, 0.9), (1000, 1000, 0, 0.5),
                                           (4096, 32000, 0, -1.0)])
@pytest.mark.parametrize("fast", [1, 0])
@with_diagnostics  # (forces the general kernel and reads the launch description: diagnostics build)
def test_int8_gemv_vs_truth(oracle, Z, N, dtype, tau, fast):
    """fast = 1: the lean kernel's int8 instantiations where the shape qualifies (whole 128-column tiles, whole chunks);
    fast = 0: the general kernel everywhere.  Same lane <-> row mapping and arithmetic order: identical bits."""
    from teal_amd import _lib
    L = _lib.load()
    L.teal_set_fast(fast)
    try:
        _int8_gemv_vs_truth(oracle, Z, N, dtype, tau, L, fast)
    finally:
        L.teal_set_fast(1)


_INT8_SEEN = {}


def _int8_gemv_vs_truth(oracle, Z, N, dtype, tau, L, fast):
    q = hash_int8(oracle, N, Z, 91)
    xb = oracle.hash_uniform(Z, 92, 2.0, dtype)
    sc_f = (0.5 + np.arange(N) % 7) * 1e-3
    scb = oracle.to_bits(sc_f.astype(np.float32), dtype)
    x = torch_from_bits(xb, dtype, DEV).view(1, 1, Z)
    W = colmajor_int8(q, 128)
    y = K().splitk_sparse_gemv_int8(x, W, torch_from_bits(scb, dtype, DEV), tau if tau > 0 else float("-inf"))
    got = oracle.from_bits(bits_from_torch(y.view(-1)), dtype)
    truth = oracle.int8_truth64(xb, q, scb, tau, dtype=dtype)
    bad = np.abs(got - truth) > tolerance(oracle, truth, dtype)
    assert not bad.any(), (int(bad.sum()), float(np.abs(got - truth).max()))
    desc = L.teal_last_launch_desc().decode()
    is_lean = "gemv_fast_kernel" in desc and ",4,true," in desc
    assert not (is_lean and not fast), desc
    if fast and (Z, N) in ((4096, 4096), (4096, 11008), (11008, 4096), (4096, 14336), (4096, 32000)):  # the Llama projection shapes
        assert is_lean, desc
    # deterministic
    y2 = K().splitk_sparse_gemv_int8(x, W, torch_from_bits(scb, dtype, DEV), tau if tau > 0 else float("-inf"))
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16))
    # lean and general kernels agree bit for bit (the parametrisation runs fast = 1 first)
    key = (Z, N, dtype, tau)
    if key in _INT8_SEEN:
        assert torch.equal(_INT8_SEEN.pop(key), y.view(torch.int16).cpu()), (key, fast)
    else:
        _INT8_SEEN[key] = y.view(torch.int16).cpu()


@with_diagnostics
def test_int8_qkv_three_thresholds_and_geometries(oracle):
    from teal_amd import _lib
    L = _lib.load()
    Z, N, kv, dtype = 2048, 3072, 512, 0
    q = hash_int8(oracle, N, Z, 93)
    xb = oracle.hash_uniform(Z, 94, 2.0, dtype)
    scb = oracle.to_bits(np.full(N, 2e-3, dtype=np.float32), dtype)
    x = torch_from_bits(xb, dtype, DEV).view(1, 1, Z)
    W = colmajor_int8(q, 128)
    sc = torch_from_bits(scb, dtype, DEV)
    truth = oracle.int8_truth64(xb, q, scb, 0.6, 0.3, 0.8, N - 2 * kv, kv, dtype)
    try:
        for lpr in (0, 8, 16, 32):
            for split in (0, 1, 2, 5):
                for wl in (1, 0):
                    assert L.teal_set_tuning(lpr, 0, split, 0) == 0
                    L.teal_set_wave_local(wl)
                    y = K().qkv_gemv_int8(x, W, sc, 0.6, 0.3, 0.8, 0, kv)
                    got = oracle.from_bits(bits_from_torch(y.view(-1)), dtype)
                    assert np.all(np.abs(got - truth) <= tolerance(oracle, truth, dtype)), (lpr, split, wl)
    finally:
        L.teal_set_tuning(0, 0, 0, 0)
        L.teal_set_wave_local(1)


def test_int8_error_behaviour():
    Z, N = 256, 128
    x = torch.zeros(1, 1, Z, device=DEV, dtype=torch.float16)
    W = torch.zeros(Z, N, dtype=torch.int8, device=DEV).T
    sc = torch.ones(N, device=DEV, dtype=torch.float16)
    K().splitk_sparse_gemv_int8(x, W, sc, 0.1)
    with pytest.raises(TypeError):
        K().splitk_sparse_gemv_int8(x, W, sc.to(torch.bfloat16), 0.1)
    with pytest.raises(AssertionError):
        K().splitk_sparse_gemv_int8(x, W.contiguous(), sc, 0.1)      # row-major weight: the reference's assert (sparse_gemv.py:106)
    with pytest.raises(RuntimeError):
        K().splitk_sparse_gemv_int8(x.cpu(), W, sc, 0.1)             # no CPU fallback
    with pytest.raises(RuntimeError):
        K().splitk_sparse_gemv_int8(torch.zeros(1, 2, Z, device=DEV, dtype=torch.float16), W, sc, 0.1)


def test_quantiser_matches_oracle_on_gpu(oracle):
    from teal_amd.quantize import quantize_per_channel
    g = torch.Generator(device=DEV).manual_seed(5)
    w = (torch.randn(200, 300, device=DEV, generator=g) * 0.05).half()
    w[4] = 0
    q, s = quantize_per_channel(w)
    qn, sn = oracle.quantize_per_channel_np(w.float().cpu().numpy())
    # torch's fp32 division on the GPU is not the CPU's correctly-rounded one: a code near a .5 tie can move by one
    d = np.abs(q.cpu().numpy().astype(np.int32) - qn.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 1e-2
    assert np.allclose(s.cpu().numpy(), sn, rtol=2e-7, atol=0)
    qc, sc = quantize_per_channel(w.cpu())  # same arithmetic on the CPU: bit-exact
    assert np.array_equal(qc.numpy(), qn) and np.array_equal(sc.numpy().view(np.uint32), sn.view(np.uint32))


@pytest.mark.parametrize("dtype,sparsity", [(torch.float16, 0.0), (torch.bfloat16, 0.0), (torch.float16, 0.5)])
def test_int8_engine_matches_int8_module_path(dtype, sparsity):
    """quantize_model_int8 -> monkeypatched int8 ops (eager) vs the fused engine on the same int8 weights."""
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine
    from teal_amd.quantize import quantize_model_int8
    ref = quantize_model_int8(G.build_synthetic_model("tiny-test", DEV, dtype, seed=3, std=0.05))
    ref.fused_decode = False  # op-by-op module path
    eng_m = quantize_model_int8(G.build_synthetic_model("tiny-test", DEV, dtype, seed=3, std=0.05))
    ths = G.apply_sparsity(ref, sparsity=sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
    G.apply_sparsity(eng_m, sparsity=sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
    assert ref.layers[0].attention.wqkv.weight.dtype == torch.int8 and ref.layers[0].attention.int8
    prompt = torch.tensor([5, 17, 301, 44, 9], device=DEV, dtype=torch.int)
    with torch.no_grad():
        for m in (ref, eng_m):
            m.max_seq_length = -1
            m.setup_caches(1, 64)
            m(prompt.view(1, -1), torch.arange(5, device=DEV))
        eng = DecodeEngine(eng_m, ths)
        assert eng.int8
        for step, tok_id in enumerate((7, 100, 3)):
            tok = torch.tensor([[tok_id]], device=DEV, dtype=torch.int)
            pos = torch.tensor([5 + step], device=DEV, dtype=torch.int)
            a, b = ref(tok, pos).float().view(-1), eng(tok, pos).float().view(-1)
            if sparsity == 0.0:
                tol = 6e-3 if dtype == torch.float16 else 5e-2
                assert torch.allclose(a, b, atol=tol, rtol=tol), (step, float((a - b).abs().max()))
            else:
                assert torch.nn.functional.cosine_similarity(a, b, dim=0) > 0.98


@pytest.mark.parametrize("name,tdt,sparsity", [("7B", torch.float16, 0.5), ("llama-3-8b", torch.bfloat16, 0.4)])
@with_diagnostics
def test_int8_engine_runs_lean_kernel_and_equals_general_at_real_width(name, tdt, sparsity):
    """The fused int8 engine at real widths under sparsity: every GEMV launch of a decode step is a lean-kernel int8
    instantiation (RMSNorm / attention-merge / silu*up producers, 128-column tiles; qkv and down row-sliced into slabs), and
    the step is bit-identical to the same step on the general kernel, which tests above pin against the double-precision
    truth and the reference module's fixture (gpt-fast/quantize.py:339-357)."""
    from teal_amd import _lib
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine
    from teal_amd.quantize import quantize_model_int8
    L = _lib.load()
    model = quantize_model_int8(G.build_synthetic_model(name, DEV, tdt, seed=17, n_layer=2))
    torch.cuda.empty_cache()
    ths = G.apply_sparsity(model, sparsity=sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
    prompt = torch.randint(0, model.config.vocab_size, (6,), device=DEV, dtype=torch.int, generator=torch.Generator(device=DEV).manual_seed(3))
    try:
        with torch.no_grad():
            model.max_seq_length = -1
            model.setup_caches(1, 32)
            model(prompt.view(1, -1), torch.arange(6, device=DEV))
            eng = DecodeEngine(model, ths)
            assert eng.int8 and not eng.pair
            tok = torch.tensor([[29]], device=DEV, dtype=torch.int)
            pos = torch.tensor([6], device=DEV, dtype=torch.int)
            descs = {}

            def hook(when, stage, i):
                if when == "after" and stage != "attn":
                    descs[(stage, i)] = L.teal_last_launch_desc().decode()

            bufs = lambda: [b.clone() for b in (eng.s_qkv, eng.att_ws, eng.s_wo, eng.gu, eng.s_down, eng.resid[0], eng.resid[1], eng.logits)]  # noqa: E731
            L.teal_set_fast(1)
            eng(tok, pos, hook=hook)
            lean = bufs()
            assert len(descs) == 2 * 4 + 1
            for k, d in descs.items():
                if k[0] == "head" and model.config.vocab_size > 64 * 1024:
                    continue  # a 128 k-entry vocabulary takes 256-column tiles: the general kernel
                assert "gemv_fast_kernel" in d and ",4,true," in d, (k, d)
            kept = eng.kept_fractions(tok, pos)
            assert all(0.2 < v < 0.85 for v in kept.values()), kept
            L.teal_set_fast(0)
            eng(tok, pos, hook=hook)
            assert all("sparse_gemv_kernel<" in d for d in descs.values()), descs
            for nm, a, b in zip(("s_qkv", "att_ws", "s_wo", "gate|up", "s_down", "resid A", "resid B", "logits"), lean, bufs()):
                assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), nm
    finally:
        L.teal_set_fast(1)
        del model
        torch.cuda.empty_cache()
