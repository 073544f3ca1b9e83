"""GPU: the hand-fused dense prompt pass for short prompts (teal_amd/csrc/teal_prefill.hip, teal_amd/gpt_fast/prefill.py).

Reference semantics: the prefill branch of the ops is a dense matmul (kernels/sparse_gemv.py:271,298) inside the stock gpt-fast
forward (gpt-fast/model.py:107-121,158-186,258-259,289-291).  Floating-point kernels, so:
  * the GEMM launch against oracle.truth64 with every row kept, token by token (SURVEY 8(c) tolerance: 1e-3 * max(1, |truth|) +
    one output ulp), through the C ABI, at the projections' real shapes incl. the two-image gate | up launch, T = 1, 6, 8 (one
    16-byte word per feature) and 9, 12, 16 (two);
  * the whole pass against the module path (the torch fp32-accumulating reference of the same ops): last-token logits within a
    few output ulps of their scale, the KV rows of every layer, the first sampled token of generate().
"""
import ctypes

import numpy as np
import pytest
import torch

from helpers import bits_from_torch, tolerance, torch_from_bits

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _image(bits, Z, N, dtype, pad=64):
    buf = torch.zeros(Z, N + pad, dtype=torch.float16 if dtype == 0 else torch.bfloat16, device=DEV)
    buf[:, :N] = torch_from_bits(bits, dtype, DEV).view(Z, N)
    return buf


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("Z,n0,n1", [(4096, 4096, 0), (4096, 12288, 0), (4096, 11008, 11008), (11008, 4096, 0), (8192, 10240, 0), (512, 768, 0)])
def test_prefill_gemm_vs_oracle(oracle, Z, n0, n1, dtype):
    from teal_amd import _lib, runtime
    from teal_amd.gpt_fast.prefill import IN_XT, PrefillIn
    O = oracle
    L = _lib.load()
    runtime.init()
    w0b = O.hash_uniform_c(Z * n0, 41 + n0, 0.05, dtype)
    w1b = O.hash_uniform_c(Z * n1, 43 + n1, 0.05, dtype) if n1 else None
    W0 = _image(w0b, Z, n0, dtype)
    W1 = _image(w1b, Z, n1, dtype) if n1 else None
    ntot = n0 + n1
    slabs = torch.full((16 * ntot * 16,), float("nan"), device=DEV, dtype=torch.float32)
    for T in (1, 6, 8, 9, 12, 16):
        R = 8 if T <= 8 else 16  # tokens per feature row of the hand-over layout
        xs = [O.hash_uniform(Z, 500 + 10 * T + s, 2.0, dtype) for s in range(T)]
        xt = torch.zeros(Z, R, device=DEV, dtype=W0.dtype)
        for s in range(T):
            xt[:, s] = torch_from_bits(xs[s], dtype, DEV)
        split = ctypes.c_int(0)
        gin = PrefillIn(mode=IN_XT, xt=xt.data_ptr())
        rc = L.teal_prefill_gemm(ctypes.byref(gin), W0.data_ptr(), W0.stride(0), n0, W1.data_ptr() if n1 else None, W1.stride(0) if n1 else 0, n1,
                                 slabs.data_ptr(), slabs.numel() * 4, Z, T, dtype, ctypes.byref(split), runtime.stream_ptr())
        assert rc == 0 and 1 <= split.value <= 16
        torch.cuda.synchronize()
        v = slabs[: split.value * ntot * R].view(split.value, ntot, R)
        acc = torch.zeros(ntot, R, device=DEV, dtype=torch.float32)
        for k in range(split.value):  # slice order, as the consumers sum
            acc = acc + v[k]
        got = O.from_bits(O.to_bits(acc[:, :T].T.contiguous().cpu().numpy().reshape(-1), dtype), dtype).reshape(T, ntot)
        for s in range(T):
            truth = np.concatenate([O.truth64(xs[s], w0b, Z, n0, -1.0, dtype=dtype)] +
                                   ([O.truth64(xs[s], w1b, Z, n1, -1.0, dtype=dtype)] if n1 else []))
            err = np.abs(got[s] - truth)
            assert (err <= tolerance(O, truth, dtype)).all(), (Z, n0, n1, dtype, T, s, float(err.max()))
    # argument checks: T out of range, ragged Z, a slab buffer too small
    bad = ctypes.c_int(0)
    assert L.teal_prefill_gemm(ctypes.byref(gin), W0.data_ptr(), W0.stride(0), n0, None, 0, 0, slabs.data_ptr(), slabs.numel() * 4, Z, 17, dtype,
                               ctypes.byref(bad), runtime.stream_ptr()) == -3
    assert L.teal_prefill_gemm(ctypes.byref(gin), W0.data_ptr(), W0.stride(0), n0, None, 0, 0, slabs.data_ptr(), slabs.numel() * 4, Z - 64, 6, dtype,
                               ctypes.byref(bad), runtime.stream_ptr()) == -3
    assert L.teal_prefill_gemm(ctypes.byref(gin), W0.data_ptr(), W0.stride(0), n0, None, 0, 0, slabs.data_ptr(), 64, Z, 6, dtype,
                               ctypes.byref(bad), runtime.stream_ptr()) == -5


@pytest.mark.parametrize("arch,tdt,n_layer,T", [("7B", torch.float16, 2, 6), ("7B", torch.float16, 2, 2), ("7B", torch.float16, 2, 8),
                                                 ("7B", torch.float16, 2, 9), ("7B", torch.float16, 2, 12), ("7B", torch.float16, 2, 16),
                                                 ("llama-3-8b", torch.bfloat16, 2, 6), ("llama-3-8b", torch.bfloat16, 2, 13),
                                                 ("tiny-gqa-test", torch.float16, 2, 5), ("tiny-gqa-test", torch.float16, 2, 11),
                                                 ("70B", torch.float16, 1, 6), ("70B", torch.float16, 1, 16)])
def test_fused_prompt_pass_equals_module_path(arch, tdt, n_layer, T):
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.prefill import FusedPrefill
    model = G.build_synthetic_model(arch, DEV, tdt, seed=21, n_layer=n_layer)
    G.apply_sparsity(model, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True, decode_calibration=False)
    V = model.config.vocab_size
    prompt = torch.randint(0, V, (T,), device=DEV, dtype=torch.int, generator=torch.Generator(device=DEV).manual_seed(5))
    try:
        with torch.no_grad():
            model.max_seq_length = -1
            model.setup_caches(1, 32)
            G.relayout_for_engine(model)
            want = model(prompt.view(1, -1), torch.arange(T, device=DEV))[0, -1].float().clone()  # module path: dense matmuls, torch glue
            kv_want = [(l.attention.kv_cache.k_cache[0, :, :T].float().clone(), l.attention.kv_cache.v_cache[0, :, :T].float().clone())
                       for l in model.layers]
            for l in model.layers:
                l.attention.kv_cache.k_cache.zero_()
                l.attention.kv_cache.v_cache.zero_()
            pre_e, pre_g = FusedPrefill(model, graph=False), FusedPrefill(model, graph=True)
            for pre, graph in ((pre_e, False), (pre_g, True), (pre_g, True)):  # eager, capture + replay, replay
                got = pre(prompt)
                assert pre.used == "hip" and got.shape == (1, 1, V)
                torch.cuda.synchronize()
                g = got.view(-1).float()
                scale = float(want.abs().max())
                ulp = scale * (2.0 ** -10 if tdt == torch.float16 else 2.0 ** -7)
                assert float((g - want).abs().max()) <= 6 * ulp, (arch, T, graph, float((g - want).abs().max()), ulp)
                assert float(torch.nn.functional.cosine_similarity(g, want, dim=0)) > 0.9995
                for l, (kw, vw) in zip(model.layers, kv_want):
                    tol = 2e-2 if tdt == torch.float16 else 1e-1
                    assert torch.allclose(l.attention.kv_cache.k_cache[0, :, :T].float(), kw, atol=tol, rtol=tol)
                    assert torch.allclose(l.attention.kv_cache.v_cache[0, :, :T].float(), vw, atol=tol, rtol=tol)
                    assert not l.attention.kv_cache.k_cache[0, :, T:].any(), "rows past the prompt are not touched"
            # a longer prompt takes the fallback
            long_prompt = torch.randint(0, V, (17,), device=DEV, dtype=torch.int)
            out = FusedPrefill(model, graph=False)
            y = out(long_prompt)
            assert out.used == "fallback" and y.shape == (1, 17, V)
            # ... and so does a one-token prompt: a decode step in the reference too (its ops run the sparse kernel at S == 1)
            out(long_prompt[:1])
            assert out.used == "fallback"
    finally:
        del model
        torch.cuda.empty_cache()


def test_generate_uses_the_fused_prompt_pass(capsys):
    """generate.main --compile: the prompt pass of the 6-token prompt is the HIP pass; --module_prefill keeps the patched modules;
    both decode the same number of tokens and report tokens/sec."""
    from teal_amd.gpt_fast import generate as G
    base = ["--synthetic", "7B", "--n_layer", "2", "--sparsity", "0.5", "--compile", "--num_samples", "2", "--max_new_tokens", "16"]
    r1 = G.main(G.build_parser().parse_args(base))
    r2 = G.main(G.build_parser().parse_args(base + ["--module_prefill"]))
    assert len(r1["sequences"][0]) == len(r2["sequences"][0]) == 6 + 16
    assert r1["prefill"] == "FusedPrefill:hip" and r2["prefill"] == "GraphedPrefill"
