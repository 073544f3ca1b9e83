"""GPU: the fused HIP decode step (engine.py: RMSNorm/residual/silu producers folded into the GEMV
launches + the decode-attention kernel) against the unfused module path it replaces."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _models(dtype, sparsity, seed=3):
    from teal_amd.gpt_fast import generate as G
    ref = G.build_synthetic_model("tiny-test", DEV, dtype, seed=seed, std=0.05)
    ref.fused_decode = False  # the reference side of every comparison is the op-by-op module path
    eng_m = G.build_synthetic_model("tiny-test", DEV, dtype, seed=seed, std=0.05)
    ths = G.apply_sparsity(ref, sparsity=sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
    G.apply_sparsity(eng_m, sparsity=sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
    for m in (ref, eng_m):
        m.max_seq_length = -1
        m.setup_caches(1, 64)
    return ref, eng_m, ths


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_decode_attention_kernel_vs_torch(dtype):
    from teal_amd import _lib, runtime
    from teal_amd.gpt_fast.model import apply_rotary_emb, precompute_freqs_cis
    L = _lib.load()
    runtime.init()
    for n_head, n_kv, hd, pos, S in ((4, 2, 64, 0, 64), (4, 2, 64, 17, 64), (8, 8, 128, 40, 64), (8, 2, 128, 63, 64),
                                     (4, 2, 128, 300, 512), (4, 4, 64, 700, 2048), (8, 2, 128, 1500, 2048)):
        g = torch.Generator(device=DEV).manual_seed(pos + hd)
        qkv = (torch.randn((n_head + 2 * n_kv) * hd, device=DEV, generator=g) * 0.5).to(dtype)
        kc = (torch.randn(n_kv, S, hd, device=DEV, generator=g) * 0.5).to(dtype)
        vc = (torch.randn(n_kv, S, hd, device=DEV, generator=g) * 0.5).to(dtype)
        rope = precompute_freqs_cis(S, hd, 10000, dtype).to(DEV).contiguous()
        y = torch.empty(n_head * hd, device=DEV, dtype=dtype)
        p = torch.tensor([pos], device=DEV, dtype=torch.int32)
        kc0, vc0 = kc.clone(), vc.clone()
        rc = L.teal_decode_attention(qkv.data_ptr(), rope.data_ptr(), p.data_ptr(), kc.data_ptr(), vc.data_ptr(), y.data_ptr(),
                                     n_head, n_kv, hd, S, runtime.dtype_code(dtype), runtime.stream_ptr())
        assert rc == 0
        # torch reference (the module path's maths)
        q, k, v = qkv.split([n_head * hd, n_kv * hd, n_kv * hd])
        fc = rope[pos:pos + 1]
        qr = apply_rotary_emb(q.view(1, 1, n_head, hd), fc).view(n_head, hd)
        kr = apply_rotary_emb(k.view(1, 1, n_kv, hd), fc).view(n_kv, hd)
        kc0[:, pos] = kr
        vc0[:, pos] = v.view(n_kv, hd)
        assert torch.equal(kc, kc0) and torch.equal(vc, vc0), "KV-cache append differs"
        rep = n_head // n_kv
        K = kc0[:, :pos + 1].repeat_interleave(rep, dim=0).float()
        V = vc0[:, :pos + 1].repeat_interleave(rep, dim=0).float()
        sc = torch.einsum("hd,htd->ht", qr.float(), K) / hd ** 0.5
        want = torch.einsum("ht,htd->hd", torch.softmax(sc, dim=-1), V).reshape(-1)
        tol = 4e-3 if dtype == torch.float16 else 3e-2
        assert torch.allclose(y.float(), want, atol=tol, rtol=tol), float((y.float() - want).abs().max())


@pytest.mark.parametrize("pair", [True, False])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_engine_matches_module_path_dense_thresholds(dtype, pair):
    """tau = -1 (everything kept): no threshold flips, so the fused step must track the unfused
    module path to rounding for several tokens, including the KV caches it appends to."""
    from teal_amd.gpt_fast.engine import DecodeEngine
    ref, eng_m, ths = _models(dtype, 0.0)
    prompt = torch.randint(0, 512, (6,), device=DEV, dtype=torch.int)
    with torch.no_grad():
        for m in (ref, eng_m):
            m(prompt.view(1, -1), torch.arange(6, device=DEV))
        eng = DecodeEngine(eng_m, ths, pair=pair)
        assert eng.pair == pair
        tok = torch.tensor([[11]], device=DEV, dtype=torch.int)
        for step in range(5):
            pos = torch.tensor([6 + step], device=DEV, dtype=torch.int)
            a = ref(tok, pos).float().view(-1)
            b = eng(tok, pos).float().view(-1)
            tol = 6e-3 if dtype == torch.float16 else 6e-2
            assert torch.allclose(a, b, atol=tol, rtol=tol), (step, float((a - b).abs().max()))
            assert int(a.argmax()) == int(b.argmax()) or (a - b).abs().max() < tol
            tok = a.argmax().view(1, 1).to(torch.int)
        for lr, le in zip(ref.layers, eng_m.layers):
            kr, ke = lr.attention.kv_cache.k_cache[:, :, :11].float(), le.attention.kv_cache.k_cache[:, :, :11].float()
            assert torch.allclose(kr, ke, atol=tol, rtol=tol)


def test_engine_sparse_tracks_module_path():
    """50 % thresholds: a handful of activations sit within rounding of tau and may flip, so compare
    by direction of the logits rather than element-wise."""
    from teal_amd.gpt_fast.engine import DecodeEngine
    ref, eng_m, ths = _models(torch.float16, 0.5)
    prompt = torch.tensor([3, 141, 59, 26, 500, 358], device=DEV, dtype=torch.int)
    with torch.no_grad():
        for m in (ref, eng_m):
            m(prompt.view(1, -1), torch.arange(6, device=DEV))
        eng = DecodeEngine(eng_m, ths)
        tok = torch.tensor([[5]], device=DEV, dtype=torch.int)
        for step in range(4):
            pos = torch.tensor([6 + step], device=DEV, dtype=torch.int)
            a = ref(tok, pos).float().view(-1)
            b = eng(tok, pos).float().view(-1)
            cos = torch.nn.functional.cosine_similarity(a, b, dim=0)
            assert cos > 0.98, (step, float(cos))  # dim 256: a single near-threshold flip moves the logits visibly
            tok = a.argmax().view(1, 1).to(torch.int)


def test_engine_under_hipgraph_equals_eager():
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine
    _, eng_m, ths = _models(torch.float16, 0.5)
    prompt = torch.randint(0, 512, (6,), device=DEV, dtype=torch.int)
    with torch.no_grad():
        eng_m(prompt.view(1, -1), torch.arange(6, device=DEV))
        eng = DecodeEngine(eng_m, ths)
        tok = torch.tensor([[9]], device=DEV, dtype=torch.int)
        pos = torch.tensor([6], device=DEV, dtype=torch.int)
        eager = eng(tok, pos).clone()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            eng(tok, pos)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = eng(tok, pos)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), eager.view(torch.int16))
        # a different token/position through the same graph
        tok.fill_(3)
        pos.fill_(7)
        g.replay()
        torch.cuda.synchronize()
        replayed = out.clone()
        want = eng(tok, pos)
        assert torch.equal(replayed.view(torch.int16), want.view(torch.int16))


def test_fused_gemv_resid_norm_matches_torch():
    """RESID_NORM producer alone: x = RMSNorm(resid + round(sum slabs)) * w, then a dense GEMV."""
    from teal_amd import _lib, runtime
    from teal_amd.gpt_fast.engine import GemvIn, GemvOut, _out, TEAL_IN_RESID_NORM, TEAL_OUT_ROUNDED
    from teal_amd.gpt_fast.model import RMSNorm
    L = _lib.load()
    runtime.init()
    Z, N, ns = 1024, 768, 3
    dt = torch.float16
    g = torch.Generator(device=DEV).manual_seed(0)
    resid = (torch.randn(Z, device=DEV, generator=g)).to(dt)
    slabs = torch.randn(ns, Z, device=DEV, generator=g) * 0.3
    nw = (1 + 0.1 * torch.randn(Z, device=DEV, generator=g)).to(dt)
    W = (torch.randn(N, Z, device=DEV, generator=g) * 0.05).to(dt).T.contiguous().T
    rout = torch.zeros(Z, device=DEV, dtype=dt)
    y = torch.zeros(N, device=DEV, dtype=dt)
    ws = runtime.reserve_workspace(Z, N)
    gin = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=resid.data_ptr(), slabs=slabs.data_ptr(), nslabs=ns,
                 norm_weight=nw.data_ptr(), eps=1e-5, resid_out=rout.data_ptr())
    gout = _out([(W.data_ptr(), N, 0, N, -1.0, y.data_ptr())], TEAL_OUT_ROUNDED)
    rc = L.teal_fused_gemv(ctypes.byref(gin), ctypes.byref(gout), Z, 0, ws.data_ptr(), ws.numel() * 4, None, runtime.stream_ptr())
    assert rc == 0
    # residual stream: bit-exact — the slabs are summed in slice order in fp32 and rounded once, then added to the residual
    acc = slabs[0].clone()
    for i in range(1, ns):
        acc = acc + slabs[i]
    h = (resid.float() + acc.to(dt).float()).to(dt)
    assert torch.equal(rout, h), float((rout.float() - h.float()).abs().max())
    norm = RMSNorm(Z, 1e-5).to(DEV)
    norm.weight.data = nw
    x = norm(rout.view(1, 1, Z))
    want = torch.matmul(x.float(), W.float().T).view(-1)
    assert torch.allclose(y.float(), want, atol=3e-3, rtol=3e-3), float((y.float() - want).abs().max())


def test_fused_sampler_topk_and_distribution():
    """top-k pivot is exact (ties kept), temperature -> 0 is argmax, and draws follow softmax(top-k)."""
    from teal_amd import _lib, runtime
    L = _lib.load()
    runtime.init()
    V = 4096
    g = torch.Generator(device=DEV).manual_seed(5)
    logits = (torch.randn(V, device=DEV, generator=g) * 2).to(torch.float16)
    state = torch.tensor([77, 0], dtype=torch.int64, device=DEV)
    tok = torch.zeros(1, dtype=torch.int32, device=DEV)

    def draw(n, top_k, temp):
        out = []
        for _ in range(n):
            rc = L.teal_sample_topk(logits.data_ptr(), V, 0, top_k, temp, state.data_ptr(), tok.data_ptr(), None, None, 0, runtime.stream_ptr())
            assert rc == 0
            out.append(int(tok.item()))
        return out

    assert draw(3, 50, 1e-6) == [int(logits.float().argmax())] * 3      # T -> 0: argmax
    assert int(state[1]) == 3                                            # counter bumped per draw
    k = 8
    top = set(torch.topk(logits.float(), k).indices.tolist())
    draws = draw(400, k, 1.0)
    assert set(draws) <= top and len(set(draws)) >= 4                    # only top-k, and it does vary
    # empirical frequencies ~ softmax over the top-k
    idx = torch.topk(logits.float(), k).indices
    p = torch.softmax(logits.float()[idx], dim=0).cpu().numpy()
    freq = np.array([draws.count(int(i)) for i in idx]) / len(draws)
    assert np.abs(freq - p).max() < 0.08
    assert len(set(draw(200, 0, 1.0))) > 20                              # no filter: wide support


def test_engine_device_resident_loop_matches_stepwise():
    """decode_n (one graph replay per token, token/position/history carried on the device) produces
    the same tokens as stepping the engine and the sampler by hand with the same RNG state."""
    from teal_amd.gpt_fast.engine import DecodeEngine
    _, eng_m, ths = _models(torch.float16, 0.5)
    prompt = torch.randint(0, 512, (6,), device=DEV, dtype=torch.int)
    with torch.no_grad():
        eng_m(prompt.view(1, -1), torch.arange(6, device=DEV))
        eng = DecodeEngine(eng_m, ths)
        first = torch.tensor([[9]], device=DEV, dtype=torch.int)
        toks = eng.decode_n(first, 6, 10, temperature=0.8, top_k=50, use_graph=True)
        assert toks.shape == (10,) and int(eng.pos_buf) == 16
        # by hand, eager, same seed/counter sequence
        eng.rng_state.copy_(torch.tensor([eng._seed + eng._calls, 0], dtype=torch.int64))
        cur = first.clone()
        hand = []
        for i in range(10):
            pos = torch.tensor([6 + i], device=DEV, dtype=torch.int)
            lg = eng(cur, pos)
            t = eng.sample_fused(lg, 0.8, 50)
            hand.append(int(t))
            cur = t.view(1, 1).clone()
        assert toks.tolist() == hand
        toks2 = eng.decode_n(first, 6, 10, temperature=0.8, top_k=50, use_graph=False)
        assert toks2.shape == (10,)


def test_generate_with_engine_decoder():
    from teal_amd.gpt_fast import generate as G
    m = G.build_synthetic_model("tiny-test", DEV, torch.float16, seed=3, std=0.05)
    ths = G.apply_sparsity(m, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True)
    dec = G.EngineDecoder(m, ths, True, 0.8, 50)
    prompt = torch.randint(0, 512, (6,), device=DEV, dtype=torch.int)
    y = G.generate(m, prompt, 12, dec, temperature=0.8, top_k=50)
    assert y.numel() == 18 and torch.equal(y[:6], prompt) and int(y.max()) < 512 and int(y.min()) >= 0


def test_pair_and_masked_stages_equal_unfused_stages():
    """PAIR (gate|up in one workgroup + silu*mul epilogue + masks) and MASKED consumers reproduce the
    SILU_MUL / PLAIN path: bit for bit with one threshold for gate and up (the same kept-row list, hence the same fp32
    summation order), and to the last place of h with two different thresholds — the paired workgroup streams the UNION of
    the two keep sets (teal_gemv_fast.h: tau = min(tau_gate, tau_up), rows outside a set multiply zero weights), so a gate row
    sits at another list position than in the unpaired gate-only list and its fp32 partial sums associate differently: a
    rounding of h may flip in the last place now and then (round 6: seen once the fp16 roundings stopped being folded into
    v_fma_mixlo_f16; with the folded build the seeded inputs below happened not to hit one)."""
    from teal_amd.gpt_fast.engine import DecodeEngine
    prompt = torch.tensor([3, 141, 59, 26, 500, 358], device=DEV, dtype=torch.int)
    for up_scale in (1.0, 0.8):
        _, m1, ths = _models(torch.float16, 0.5)
        _, m2, _ = _models(torch.float16, 0.5)
        for t in ths:  # (0.8: block-wise-greedy style, gate and up thresholds differ)
            t["up"] = t["gate"] * up_scale
        with torch.no_grad():
            for m in (m1, m2):
                m(prompt.view(1, -1), torch.arange(6, device=DEV))
            e1, e2 = DecodeEngine(m1, ths, pair=True), DecodeEngine(m2, ths, pair=False)
            tok = torch.tensor([[5]], device=DEV, dtype=torch.int)
            for step in range(4):
                pos = torch.tensor([6 + step], device=DEV, dtype=torch.int)
                a, b = e1(tok, pos), e2(tok, pos)
                if up_scale == 1.0:
                    assert torch.equal(a.view(torch.int16), b.view(torch.int16)), step
                else:
                    ulp = float(b.float().abs().max()) * 2.0 ** -10
                    assert float((a.float() - b.float()).abs().max()) <= 16 * ulp, (step, float((a.float() - b.float()).abs().max()), ulp)
                tok = b.float().argmax().view(1, 1).to(torch.int)


def test_interleaved_slabs_equal_planar():
    """SLABS producer with slabs_interleaved + RESID_NORM consumer reading them == planar slabs."""
    from teal_amd import _lib, runtime
    from teal_amd.gpt_fast.engine import GemvIn, _out, TEAL_IN_PLAIN, TEAL_IN_RESID_NORM, TEAL_OUT_ROUNDED, TEAL_OUT_SLABS
    L = _lib.load()
    runtime.init()
    Z, N = 4096, 4096
    dt = torch.float16
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(Z, device=DEV, generator=g).to(dt)
    W = (torch.randn(N, Z, device=DEV, generator=g) * 0.03).to(dt).T.contiguous().T
    W2 = (torch.randn(1024, N, device=DEV, generator=g) * 0.03).to(dt).T.contiguous().T
    resid = torch.randn(N, device=DEV, generator=g).to(dt)
    nw = torch.ones(N, device=DEV, dtype=dt)
    ws = runtime.reserve_workspace(Z, N)
    outs = []
    for il in (0, 1):
        slabs = torch.zeros(32 * N, device=DEV, dtype=torch.float32)
        gout = _out([(W.data_ptr(), N, 0, N, 0.7, None)], TEAL_OUT_SLABS, slabs)
        gout.slabs_interleaved = il
        ns = ctypes.c_int(0)
        gin = GemvIn(mode=TEAL_IN_PLAIN, x=x.data_ptr())
        assert L.teal_fused_gemv(ctypes.byref(gin), ctypes.byref(gout), Z, 0, ws.data_ptr(), ws.numel() * 4, ctypes.byref(ns), runtime.stream_ptr()) == 0
        assert 1 < ns.value <= 8
        y = torch.zeros(1024, device=DEV, dtype=dt)
        rout = torch.zeros(N, device=DEV, dtype=dt)
        cin = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=resid.data_ptr(), slabs=slabs.data_ptr(), nslabs=ns.value, slabs_interleaved=il,
                     norm_weight=nw.data_ptr(), eps=1e-5, resid_out=rout.data_ptr())
        cout = _out([(W2.data_ptr(), 1024, 0, 1024, 0.5, y.data_ptr())], TEAL_OUT_ROUNDED)
        assert L.teal_fused_gemv(ctypes.byref(cin), ctypes.byref(cout), N, 0, ws.data_ptr(), ws.numel() * 4, None, runtime.stream_ptr()) == 0
        outs.append((y.clone(), rout.clone()))
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16))


@pytest.mark.parametrize("name,dtype,n_layer,int8", [("7B", torch.float16, 2, False), ("llama-3-8b", torch.bfloat16, 1, False),
                                                     ("7B", torch.float16, 2, True), ("llama-3-8b", torch.bfloat16, 1, True)])
def test_engine_matches_module_path_full_width(name, dtype, n_layer, int8):
    """real layer widths (MHA 4096/11008 and GQA 4096/14336 with the 128k vocabulary), every row kept so
    that no threshold can flip: fused engine vs unfused module path over a few tokens — 16-bit and int8
    weight-only weights (the int8 launch geometry differs: 128-column tiles, sliced wqkv, no PAIR)."""
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine
    from teal_amd.quantize import quantize_model_int8
    ref = G.build_synthetic_model(name, DEV, dtype, seed=5, n_layer=n_layer)
    ref.fused_decode = False
    eng_m = G.build_synthetic_model(name, DEV, dtype, seed=5, n_layer=n_layer)
    if int8:
        quantize_model_int8(ref)
        quantize_model_int8(eng_m)
    ths = G.apply_sparsity(ref, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True)
    G.apply_sparsity(eng_m, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True)
    V = ref.config.vocab_size
    prompt = torch.randint(0, V, (6,), device=DEV, dtype=torch.int, generator=torch.Generator(device=DEV).manual_seed(2))
    with torch.no_grad():
        for m in (ref, eng_m):
            m.max_seq_length = -1
            m.setup_caches(1, 32)
            m(prompt.view(1, -1), torch.arange(6, device=DEV))
        eng = DecodeEngine(eng_m, ths)
        tok = torch.tensor([[17]], device=DEV, dtype=torch.int)
        for step in range(3):
            pos = torch.tensor([6 + step], device=DEV, dtype=torch.int)
            a = ref(tok, pos).float().view(-1)
            b = eng(tok, pos).float().view(-1)
            scale = float(a.abs().max())
            tol = (4e-3 if dtype == torch.float16 else 4e-2) * max(1.0, scale) * (2.0 if int8 else 1.0)
            assert float((a - b).abs().max()) <= tol, (step, float((a - b).abs().max()), scale)
            tok = a.argmax().view(1, 1).to(torch.int)
    del ref, eng_m, eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name,dtype,n_layer,fmt", [("13B", torch.float16, 1, "w16"), ("13B", torch.bfloat16, 1, "int4"), ("30B", torch.bfloat16, 1, "w16"),
                                                    ("30B", torch.float16, 1, "int8"), ("34B", torch.float16, 1, "int8"), ("34B", torch.float16, 1, "int4"),
                                                    ("stories110M", torch.float16, 2, "w16"), ("Mistral-7B", torch.bfloat16, 1, "int4"),
                                                    ("70B", torch.float16, 1, "int4")])
def test_engine_matches_module_path_other_architectures(name, dtype, n_layer, fmt):
    """The other architectures of the reference's table (gpt-fast/model.py:66-79: 13B 5120 / 13824, 30B 6656 / 17920 with 52
    heads, 34B GQA 8192 / 22016, stories110M head_dim 64, Mistral-7B) and the three weight formats: the fused engine against the
    op-by-op module path with every row kept (no threshold can flip) over a few tokens, then a step at 50 % (thresholds
    taken on the decode activations)."""
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine, pick_engine
    from teal_amd.quantize import quantize_model_int4, quantize_model_int8
    quant = {"w16": lambda m: m, "int8": quantize_model_int8, "int4": lambda m: quantize_model_int4(m, 32)}[fmt]
    ref = quant(G.build_synthetic_model(name, DEV, dtype, seed=5, n_layer=n_layer))
    ref.fused_decode = False
    eng_m = quant(G.build_synthetic_model(name, DEV, dtype, seed=5, n_layer=n_layer))
    ths = G.apply_sparsity(ref, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True)
    G.apply_sparsity(eng_m, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True)
    V = ref.config.vocab_size
    prompt = torch.randint(0, V, (6,), device=DEV, dtype=torch.int, generator=torch.Generator(device=DEV).manual_seed(2))
    with torch.no_grad():
        for m in (ref, eng_m):
            m.max_seq_length = -1
            m.setup_caches(1, 32)
            m(prompt.view(1, -1), torch.arange(6, device=DEV))
        assert pick_engine(eng_m) == (DecodeEngine, None)
        eng = DecodeEngine(eng_m, ths)
        assert eng.int4 == (fmt == "int4") and eng.int8 == (fmt == "int8")
        tok = torch.tensor([[17]], device=DEV, dtype=torch.int)
        for step in range(3):
            pos = torch.tensor([6 + step], device=DEV, dtype=torch.int)
            a = ref(tok, pos).float().view(-1)
            b = eng(tok, pos).float().view(-1)
            scale = float(a.abs().max())
            tol = (4e-3 if dtype == torch.float16 else 4e-2) * max(1.0, scale) * (1.0 if fmt == "w16" else 2.0)
            assert float((a - b).abs().max()) <= tol, (step, float((a - b).abs().max()), scale)
            tok = a.argmax().view(1, 1).to(torch.int)
        # 50 %: thresholds from the decode activations themselves; the step keeps about half of every projection's rows
        sp = {p: [0.5] * n_layer for p in eng.SITE}
        eng.calibrate_on_decode(sp, torch.tensor([17], device=DEV, dtype=torch.int), 9, 6, n_samples=3, rounds=2)
        tok, pos = torch.tensor([[17]], device=DEV, dtype=torch.int), torch.tensor([12], device=DEV, dtype=torch.int)
        kept = eng.kept_fractions(tok, pos)
        assert all(0.3 < v < 0.7 for v in kept.values()), kept
        assert bool(torch.isfinite(eng(tok, pos).float()).all())
    del ref, eng_m, eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_decode_attention_split_long_context(dtype):
    """flash-decoding form: same result as the single-workgroup kernel and as torch, at thousands of positions"""
    from teal_amd import _lib, runtime
    from teal_amd.gpt_fast.model import apply_rotary_emb, precompute_freqs_cis
    L = _lib.load()
    runtime.init()
    code = runtime.dtype_code(dtype)
    # n_head / n_kv of 4 or 8 runs the grouped-query kernel (one workgroup per KV head and split, all its query heads)
    for n_head, n_kv, hd, pos, S, nsplit in ((8, 2, 128, 5000, 8192, 8), (4, 4, 64, 3, 4096, 16), (8, 8, 128, 4095, 4096, 5),
                                             (64, 8, 128, 16000, 16384, 32), (16, 2, 64, 37, 1024, 4), (32, 8, 128, 226, 464, 4),
                                             (8, 1, 128, 0, 512, 8), (16, 4, 64, 1023, 1024, 3), (6, 2, 128, 700, 2048, 4),
                                             (16, 2, 64, 4000, 4096, 16), (16, 4, 64, 37, 4096, 8), (8, 1, 128, 0, 8192, 32),
                                             (32, 8, 128, 8191, 8192, 64), (16, 2, 64, 2500, 3072, 16), (64, 8, 128, 2047, 2048, 32)):
        g = torch.Generator(device=DEV).manual_seed(pos + hd)
        qkv = (torch.randn((n_head + 2 * n_kv) * hd, device=DEV, generator=g) * 0.5).to(dtype)
        kc = (torch.randn(n_kv, S, hd, device=DEV, generator=g) * 0.5).to(dtype)
        vc = (torch.randn(n_kv, S, hd, device=DEV, generator=g) * 0.5).to(dtype)
        rope = precompute_freqs_cis(S, hd, 10000, dtype).to(DEV).contiguous()
        p = torch.tensor([pos], device=DEV, dtype=torch.int32)
        kc1, vc1, kc2, vc2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
        y1 = torch.empty(n_head * hd, device=DEV, dtype=dtype)
        y2 = torch.empty_like(y1)
        m1 = torch.zeros(n_head * hd // 64, device=DEV, dtype=torch.int64)
        m2 = torch.zeros_like(m1)
        ws = torch.zeros(n_head * nsplit * (hd + 2), device=DEV, dtype=torch.float32)
        single = S <= 8192  # the single-workgroup kernel keeps every score in LDS
        if single:
            assert L.teal_decode_attention_masked(qkv.data_ptr(), rope.data_ptr(), p.data_ptr(), kc1.data_ptr(), vc1.data_ptr(),
                                                  y1.data_ptr(), m1.data_ptr(), 0.02, n_head, n_kv, hd, S, code, runtime.stream_ptr()) == 0
        assert L.teal_decode_attention_split(qkv.data_ptr(), rope.data_ptr(), p.data_ptr(), kc2.data_ptr(), vc2.data_ptr(), y2.data_ptr(),
                                             m2.data_ptr(), 0.02, n_head, n_kv, hd, S, nsplit, ws.data_ptr(), ws.numel() * 4, code,
                                             runtime.stream_ptr()) == 0
        tol = 4e-3 if dtype == torch.float16 else 3e-2
        if single:
            assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
            assert torch.allclose(y1.float(), y2.float(), atol=tol, rtol=tol), float((y1.float() - y2.float()).abs().max())
        q, k, v = qkv.split([n_head * hd, n_kv * hd, n_kv * hd])
        qr = apply_rotary_emb(q.view(1, 1, n_head, hd), rope[pos:pos + 1]).view(n_head, hd)
        rep = n_head // n_kv
        K = kc2[:, :pos + 1].repeat_interleave(rep, dim=0).float()
        V = vc2[:, :pos + 1].repeat_interleave(rep, dim=0).float()
        want = torch.einsum("ht,htd->hd", torch.softmax(torch.einsum("hd,htd->ht", qr.float(), K) / hd ** 0.5, dim=-1), V).reshape(-1)
        assert torch.allclose(y2.float(), want, atol=tol, rtol=tol)
        # masks = the keep rule on the rounded output
        bits = (y2.float().abs() > 0.02).view(-1, 64)
        got = torch.stack([(m2 >> i) & 1 for i in range(64)], dim=1).bool()
        assert torch.equal(bits, got)
        # the general entry point (teal_decode_attention_split_ws; its workspace arguments are accepted and unused since round 5):
        # the same y, masks and cache rows as teal_decode_attention_split
        wsp = runtime.reserve_workspace(64, 64)
        kc3, vc3 = kc.clone(), vc.clone()
        y3, m3 = torch.zeros_like(y1), torch.zeros_like(m1)
        ws3 = torch.full_like(ws, float("nan"))
        assert L.teal_decode_attention_split_ws(qkv.data_ptr(), None, 0, rope.data_ptr(), p.data_ptr(), kc3.data_ptr(), vc3.data_ptr(),
                                                y3.data_ptr(), m3.data_ptr(), 0.02, n_head, n_kv, hd, S, nsplit, ws3.data_ptr(),
                                                ws3.numel() * 4, code, wsp.data_ptr(), wsp.numel() * 4, runtime.stream_ptr()) == 0
        assert torch.equal(y3.view(torch.int16), y2.view(torch.int16)) and torch.equal(m3, m2), (n_head, n_kv, hd, pos, nsplit)
        assert torch.equal(kc3, kc2) and torch.equal(vc3, vc2)


@pytest.mark.parametrize("name,block,plen,fused", [("tiny-test", 4096, 3000, True), ("tiny-test", 8192, 5000, False),
                                                   ("tiny-gqa-test", 4096, 3000, False), ("tiny-gqa-test", 2048, 1500, True)])
def test_engine_long_context_uses_split_attention(name, block, plen, fused):
    """8 split-KV partials merged by the wo launch up to 4096 positions; 16 + the merge launch beyond.  Grouped-query
    models (4 or 8 query heads per KV head) take the grouped kernel from 4096 cache positions (2048 with 8 heads per KV head): ~one workgroup per CU
    (n_kv x splits), merged by the merge launch."""
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine
    ref = G.build_synthetic_model(name, DEV, torch.float16, seed=3, std=0.05)
    ref.fused_decode = False
    eng_m = G.build_synthetic_model(name, DEV, torch.float16, seed=3, std=0.05)
    for m in (ref, eng_m):
        m.config.block_size = block
    ths = G.apply_sparsity(ref, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True)
    G.apply_sparsity(eng_m, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True)
    prompt = torch.randint(0, 512, (plen,), device=DEV, dtype=torch.int, generator=torch.Generator(device=DEV).manual_seed(4))
    with torch.no_grad():
        for m in (ref, eng_m):
            m.max_seq_length = -1
            m.setup_caches(1, block)
            m(prompt.view(1, -1), torch.arange(plen, device=DEV))
        eng = DecodeEngine(eng_m, ths)
        assert eng.att_split >= 2 and eng.att_fused_merge == fused
        if name == "tiny-gqa-test" and block >= 4096:
            assert eng.att_split == 64
        tok = torch.tensor([[7]], device=DEV, dtype=torch.int)
        pos = torch.tensor([plen], device=DEV, dtype=torch.int)
        a, b = ref(tok, pos).float().view(-1), eng(tok, pos).float().view(-1)
        assert torch.allclose(a, b, atol=8e-3, rtol=8e-3), float((a - b).abs().max())


@pytest.mark.parametrize("dtype,n_head,hd,pos,nsplit", [(torch.float16, 32, 128, 200, 4), (torch.bfloat16, 8, 64, 5, 4),
                                                        (torch.float16, 64, 128, 1900, 4), (torch.float16, 16, 64, 0, 4),
                                                        (torch.float16, 32, 128, 1500, 8), (torch.bfloat16, 64, 128, 3, 8),
                                                        (torch.float16, 16, 64, 2047, 8)])
def test_attn_merge_producer_equals_merge_launch(dtype, n_head, hd, pos, nsplit):
    """ATTN_MERGE: the wo launch merges the 4 split-KV partials itself; same projection as merge launch + GEMV."""
    from teal_amd import _lib, runtime
    from teal_amd.gpt_fast.engine import GemvIn, _out, TEAL_IN_ATTN_MERGE, TEAL_OUT_ROUNDED
    from teal_amd.gpt_fast.model import precompute_freqs_cis
    L = _lib.load()
    runtime.init()
    code = runtime.dtype_code(dtype)
    n_kv, S = n_head // 4, 2048
    Z = n_head * hd
    N = 512
    g = torch.Generator(device=DEV).manual_seed(pos + hd)
    qkv = (torch.randn((n_head + 2 * n_kv) * hd, device=DEV, generator=g) * 0.5).to(dtype)
    kc = (torch.randn(n_kv, S, hd, device=DEV, generator=g) * 0.5).to(dtype)
    vc = (torch.randn(n_kv, S, hd, device=DEV, generator=g) * 0.5).to(dtype)
    rope = precompute_freqs_cis(S, hd, 10000, dtype).to(DEV).contiguous()
    p = torch.tensor([pos], device=DEV, dtype=torch.int32)
    W = (torch.randn(N, Z, device=DEV, generator=g) * 0.05).to(dtype).T.contiguous().T
    y_att = torch.empty(Z, device=DEV, dtype=dtype)
    msk = torch.zeros(Z // 64, device=DEV, dtype=torch.int64)
    ws_a = torch.zeros(n_head * nsplit * (hd + 2), device=DEV, dtype=torch.float32)
    ws_b = torch.zeros_like(ws_a)
    st = runtime.stream_ptr()
    assert L.teal_decode_attention_split(qkv.data_ptr(), rope.data_ptr(), p.data_ptr(), kc.clone().data_ptr(), vc.clone().data_ptr(),
                                         y_att.data_ptr(), msk.data_ptr(), 0.0, n_head, n_kv, hd, S, nsplit, ws_a.data_ptr(),
                                         ws_a.numel() * 4, code, st) == 0
    # partials only (y = NULL): no merge launch
    assert L.teal_decode_attention_split(qkv.data_ptr(), rope.data_ptr(), p.data_ptr(), kc.data_ptr(), vc.data_ptr(), None, None, 0.0,
                                         n_head, n_kv, hd, S, nsplit, ws_b.data_ptr(), ws_b.numel() * 4, code, st) == 0
    assert torch.equal(ws_a, ws_b)
    ws = runtime.reserve_workspace(Z, N)
    y = torch.zeros(N, device=DEV, dtype=dtype)
    gin = GemvIn(mode=TEAL_IN_ATTN_MERGE, x=ws_b.data_ptr(), att_head_dim=hd, att_nsplit=nsplit)
    gout = _out([(W.data_ptr(), N, 0, N, -1.0, y.data_ptr())], TEAL_OUT_ROUNDED)
    assert L.teal_fused_gemv(ctypes.byref(gin), ctypes.byref(gout), Z, code, ws.data_ptr(), ws.numel() * 4, None, st) == 0
    want = torch.matmul(y_att.float(), W.float().T).view(-1)
    tol = 3e-3 if dtype == torch.float16 else 2e-2
    assert torch.allclose(y.float(), want, atol=tol, rtol=tol), float((y.float() - want).abs().max())
    # with a threshold: the keep rule acts on the merged, rounded activation (elements within an ulp of the
    # threshold may differ between the two merge roundings; bound the damage instead of demanding equality)
    tau = float(y_att.float().abs().median())
    gout = _out([(W.data_ptr(), N, 0, N, tau, y.data_ptr())], TEAL_OUT_ROUNDED)
    assert L.teal_fused_gemv(ctypes.byref(gin), ctypes.byref(gout), Z, code, ws.data_ptr(), ws.numel() * 4, None, st) == 0
    xs = torch.where(y_att.float().abs() > tau, y_att.float(), torch.zeros((), device=DEV))
    want = torch.matmul(xs, W.float().T).view(-1)
    assert torch.allclose(y.float(), want, atol=4 * tol, rtol=4 * tol), float((y.float() - want).abs().max())
    # wrong head_dim is refused
    for bad in (GemvIn(mode=TEAL_IN_ATTN_MERGE, x=ws_b.data_ptr(), att_head_dim=96, att_nsplit=nsplit),
                GemvIn(mode=TEAL_IN_ATTN_MERGE, x=ws_b.data_ptr(), att_head_dim=hd, att_nsplit=5)):
        assert L.teal_fused_gemv(ctypes.byref(bad), ctypes.byref(gout), Z, code, ws.data_ptr(), ws.numel() * 4, None, st) < 0


@pytest.mark.parametrize("dtype,nslabs", [(torch.float16, 1), (torch.bfloat16, 2), (torch.float16, 5), (torch.float16, 8)])
def test_attention_from_qkv_slabs_equals_rounded_qkv(dtype, nslabs):
    """teal_decode_attention_split_slabs sums the projection's split-K slabs itself: same bits as reducing first."""
    from teal_amd import _lib, runtime
    from teal_amd.gpt_fast.model import precompute_freqs_cis
    L = _lib.load()
    runtime.init()
    code = runtime.dtype_code(dtype)
    n_head, n_kv, hd, S, nsplit, pos = 8, 2, 128, 512, 4, 77
    n = (n_head + 2 * n_kv) * hd
    g = torch.Generator(device=DEV).manual_seed(nslabs)
    stride = (nslabs + 3) & ~3
    slabs = torch.zeros(n, stride, device=DEV, dtype=torch.float32)
    slabs[:, :nslabs] = torch.randn(n, nslabs, device=DEV, generator=g) * 0.3
    slabs[:, nslabs:] = 7.0  # padding lanes of the interleaved layout must be ignored
    acc = torch.zeros(n, device=DEV, dtype=torch.float32)
    for s_ in range(nslabs):  # slice order, fp32
        acc = acc + slabs[:, s_]
    qkv = acc.to(dtype)
    kc = (torch.randn(n_kv, S, hd, device=DEV, generator=g) * 0.5).to(dtype)
    vc = (torch.randn(n_kv, S, hd, device=DEV, generator=g) * 0.5).to(dtype)
    rope = precompute_freqs_cis(S, hd, 10000, dtype).to(DEV).contiguous()
    p = torch.tensor([pos], device=DEV, dtype=torch.int32)
    kc2, vc2 = kc.clone(), vc.clone()
    ws_a = torch.zeros(n_head * nsplit * (hd + 2), device=DEV, dtype=torch.float32)
    ws_b = torch.zeros_like(ws_a)
    st = runtime.stream_ptr()
    assert L.teal_decode_attention_split(qkv.data_ptr(), rope.data_ptr(), p.data_ptr(), kc.data_ptr(), vc.data_ptr(), None, None, 0.0,
                                         n_head, n_kv, hd, S, nsplit, ws_a.data_ptr(), ws_a.numel() * 4, code, st) == 0
    assert L.teal_decode_attention_split_slabs(slabs.data_ptr(), nslabs, rope.data_ptr(), p.data_ptr(), kc2.data_ptr(), vc2.data_ptr(),
                                               None, None, 0.0, n_head, n_kv, hd, S, nsplit, ws_b.data_ptr(), ws_b.numel() * 4,
                                               code, st) == 0
    assert torch.equal(ws_a, ws_b) and torch.equal(kc, kc2) and torch.equal(vc, vc2)
    assert L.teal_decode_attention_split_slabs(slabs.data_ptr(), 9, rope.data_ptr(), p.data_ptr(), kc2.data_ptr(), vc2.data_ptr(),
                                               None, None, 0.0, n_head, n_kv, hd, S, nsplit, ws_b.data_ptr(), ws_b.numel() * 4,
                                               code, st) < 0


@pytest.mark.parametrize("dtype,V,law", [(torch.float16, 32000, "normal"), (torch.bfloat16, 128256, "normal"), (torch.float16, 50304, "normal"),
                                         (torch.bfloat16, 4096, "normal"), (torch.float16, 32000, "flat"), (torch.bfloat16, 32000, "flat"),
                                         (torch.float16, 32000, "spike"), (torch.float16, 8200, "normal"), (torch.bfloat16, 131072, "flat"),
                                         (torch.float16, 128256, "ties")])
def test_sampler_window_and_radix_kernels_pick_the_same_tokens(dtype, V, law):
    """vocab % 8 == 0 runs the register-resident window-select kernel, V + 4 (logit -inf appended) the generic radix
    kernel: same pivot, same kept set, same counter-based random numbers -> identical tokens.  `flat` (uniform over
    +-100, large k) forces the window to widen; `spike` has one dominant logit; ties at the pivot are kept."""
    from teal_amd import _lib, runtime
    L = _lib.load()
    runtime.init()
    code = runtime.dtype_code(dtype)
    g = torch.Generator(device=DEV).manual_seed(V)
    if law == "normal":
        base = (torch.randn(V, device=DEV, generator=g) * 3).to(dtype)
        base[100:140] = base.float().max().to(dtype)          # a tie at the top: top_k = 20 must keep all of it
    elif law == "ties":  # a few distinct values: thousands of ties at every pivot (candidate lists overflow -> generic path)
        base = torch.randint(-3, 4, (V,), device=DEV, generator=g).to(dtype)
    elif law == "flat":
        base = ((torch.rand(V, device=DEV, generator=g) - 0.5) * 200).to(dtype)
    else:
        base = (torch.randn(V, device=DEV, generator=g) * 0.01).to(dtype)
        base[777] = 60.0
    padded = torch.cat([base, torch.full((4,), float("-inf"), device=DEV, dtype=dtype)])
    tok = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = runtime.reserve_workspace(64, 64)  # prepared (teal_workspace_init): its header holds the multi-workgroup scratch
    for top_k, temp in ((200, 0.8), (20, 1.0), (1, 1.0), (0, 1.0), (5000, 2.0), (V - 1, 1.5)):
        outs = []
        # multi-workgroup kernel (8192 < V <= 131072, filter on, prepared workspace), generic radix kernel (padded vocabulary), and
        # the single-workgroup window kernel (the same logits without a workspace: teal_sample_topk)
        for logits, n, use_ws in ((base, V, True), (padded, V + 4, True), (base, V, False)):
            state = torch.tensor([1234, 0], dtype=torch.int64, device=DEV)
            pos = torch.tensor([11], dtype=torch.int32, device=DEV)
            hist = torch.full((64,), -1, dtype=torch.int32, device=DEV)
            seq = []
            for _ in range(24):
                if use_ws:
                    rc = L.teal_sample_topk_ws(logits.data_ptr(), n, code, top_k, temp, state.data_ptr(), tok.data_ptr(), pos.data_ptr(),
                                               hist.data_ptr(), 64, ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
                else:
                    rc = L.teal_sample_topk(logits.data_ptr(), n, code, top_k, temp, state.data_ptr(), tok.data_ptr(), pos.data_ptr(),
                                            hist.data_ptr(), 64, runtime.stream_ptr())
                assert rc == 0
                seq.append(int(tok.item()))
            assert int(state[1]) == 24 and int(pos[0]) == 11 + 24 and hist[:24].tolist() == seq
            outs.append(seq)
        assert outs[0] == outs[1], (top_k, outs[0][:8], outs[1][:8])
        assert outs[0] == outs[2], (top_k, outs[0][:8], outs[2][:8])
        assert max(outs[0]) < V
        if top_k == 20 and law == "normal":
            tied = set(torch.nonzero(base == base.float().max().to(dtype)).view(-1).tolist())
            assert len(tied) >= 40 and set(outs[0]) <= tied and len(set(outs[0])) > 10
        if law == "spike" and top_k in (1, 20, 200):
            assert set(outs[0]) == {777}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_decode_attention_roped_long_context(dtype):
    """teal_decode_attention_split_roped (rotated q in, the token's k / v rows already in the caches) against torch at cache
    lengths on both sides of what the blind K / V prefetch covers (4 row groups per workgroup): past it the kernel keeps 8 row
    groups of K and of V in flight and refills each register as it is consumed (round 4) — one, several and partial rounds of
    that loop, 4- and 16-wave workgroups, both head sizes, grouped heads.  Rows past the position hold NaN: a clamped or
    stray read that reached the sums would show."""
    from teal_amd import _lib, runtime
    L = _lib.load()
    runtime.init()
    code = runtime.dtype_code(dtype)
    #          heads kv  hd   pos   S    nsplit      (row groups per workgroup = ceil(ceil((pos + 1) / step) / nsplit))
    cases = ((32, 32, 128, 235, 333, 4),     # the headline's launch: 16-row groups, 4 per workgroup, no refill
             (32, 32, 128, 300, 333, 4),     # same launch one group further: 5
             (8, 8, 128, 511, 512, 4),       # 4-wave workgroups, 8 groups: one full round of the loop
             (8, 8, 128, 2047, 4096, 8),     # 16-wave workgroups, 64-row groups: exactly 4 (no refill)
             (8, 8, 128, 2048, 4096, 8),     # ... and 5 in ONE workgroup of each head only
             (8, 8, 128, 3800, 4096, 8),     # bench.py's long-context point: 8
             (16, 4, 64, 4095, 4096, 4),     # head_dim 64 (32-row groups at 4 waves do not apply: 16 waves, 128-row groups), grouped heads
             (8, 2, 128, 8191, 8192, 8),     # 16 groups: two rounds
             (4, 4, 64, 5000, 8192, 3),      # ragged: 14 groups of 128 rows over 3 workgroups (5 / 5 / 4)
             (8, 8, 128, 0, 4096, 8), (8, 1, 128, 63, 2048, 4), (8, 8, 128, 64, 2048, 4))
    for n_head, n_kv, hd, pos, S, nsplit in cases:
        g = torch.Generator(device=DEV).manual_seed(pos + hd + nsplit)
        q = (torch.randn(n_head * hd, device=DEV, generator=g) * 0.5).to(dtype)
        kc = (torch.randn(n_kv, S, hd, device=DEV, generator=g) * 0.5).to(dtype)
        vc = (torch.randn(n_kv, S, hd, device=DEV, generator=g) * 0.5).to(dtype)
        kc[:, pos + 1:] = float("nan")
        vc[:, pos + 1:] = float("nan")
        kc0, vc0 = kc.clone(), vc.clone()
        p = torch.tensor([pos], device=DEV, dtype=torch.int32)
        y = torch.empty(n_head * hd, device=DEV, dtype=dtype)
        m = torch.zeros(n_head * hd // 64, device=DEV, dtype=torch.int64)
        ws = torch.full((n_head * nsplit * (hd + 2),), float("nan"), device=DEV, dtype=torch.float32)
        assert L.teal_decode_attention_split_roped(q.data_ptr(), p.data_ptr(), kc.data_ptr(), vc.data_ptr(), y.data_ptr(), m.data_ptr(),
                                                   0.02, n_head, n_kv, hd, S, nsplit, ws.data_ptr(), ws.numel() * 4, code, None, 0,
                                                   runtime.stream_ptr()) == 0
        torch.cuda.synchronize()
        assert torch.equal(kc.view(torch.int16), kc0.view(torch.int16)) and torch.equal(vc.view(torch.int16), vc0.view(torch.int16))
        rep = n_head // n_kv
        K = kc[:, :pos + 1].repeat_interleave(rep, dim=0).float()
        V = vc[:, :pos + 1].repeat_interleave(rep, dim=0).float()
        sc = torch.einsum("hd,htd->ht", q.view(n_head, hd).float(), K) / hd ** 0.5
        want = torch.einsum("ht,htd->hd", torch.softmax(sc.to(dtype).float(), dim=-1), V).reshape(-1)
        tol = 4e-3 if dtype == torch.float16 else 3e-2
        assert torch.isfinite(y.float()).all(), (n_head, n_kv, hd, pos, S, nsplit)
        assert torch.allclose(y.float(), want, atol=tol, rtol=tol), (n_head, n_kv, hd, pos, S, nsplit, float((y.float() - want).abs().max()))
        bits = (y.float().abs() > 0.02).view(-1, 64)
        got = torch.stack([(m >> i) & 1 for i in range(64)], dim=1).bool()
        assert torch.equal(bits, got)


@pytest.mark.parametrize("name,dtype", [("7B", torch.float16), ("7B", torch.bfloat16), ("70B", torch.float16)])
def test_rope_epilogue_equals_attention_side_rope(name, dtype):
    """TEAL_OUT_QKV_ROPE (RoPE of q / the new k row and the KV-cache append in the wqkv launch's epilogue, then
    teal_decode_attention_split_roped) against the slab hand-over (the attention launch sums, rotates and appends,
    gpt-fast/model.py:170-178): the same bits in the cache rows of the token, the split-KV partials and the logits."""
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine
    model = G.build_synthetic_model(name, DEV, dtype, seed=31, n_layer=2)
    ths = G.apply_sparsity(model, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True)
    prompt = torch.randint(0, model.config.vocab_size, (9,), device=DEV, dtype=torch.int, generator=torch.Generator(device=DEV).manual_seed(6))
    try:
        with torch.no_grad():
            model.max_seq_length = -1
            model.setup_caches(1, 32)
            model(prompt.view(1, -1), torch.arange(9, device=DEV))
            eng = DecodeEngine(model, ths)
            assert eng.rope_epilogue
            tok = torch.tensor([[41]], device=DEV, dtype=torch.int)
            pos = torch.tensor([9], device=DEV, dtype=torch.int)
            caches = [(l.attention.kv_cache.k_cache, l.attention.kv_cache.v_cache) for l in model.layers]

            def run(rope_epilogue):
                eng.rope_epilogue = rope_epilogue
                eng._build(ths)
                for kc, vc in caches:  # the row of this token must be written by the step itself
                    kc[:, :, 9].zero_()
                    vc[:, :, 9].zero_()
                logits = eng(tok, pos).clone()
                return [logits, eng.att_ws.clone()] + [c[:, :, 9].clone() for kv in caches for c in kv], eng.n_qkv.value

            a, na = run(True)
            b, nb = run(False)
            if name == "70B":  # 70B-class projections keep the row-sliced slab hand-over (faster there): the request falls back
                assert na == nb and nb > 1, (na, nb)
            else:
                assert na == 0 and nb == 1, (na, nb)  # the epilogue ran / one slab was handed over: the same fp32 sums
            for i, (x, y) in enumerate(zip(a, b)):
                assert torch.equal(x.view(torch.uint8), y.view(torch.uint8)), i
            assert float(a[2].float().abs().max()) > 0  # the cache row was really written
    finally:
        del model
        torch.cuda.empty_cache()


@pytest.mark.parametrize("name,dtype", [("7B", torch.float16), ("llama-3-8b", torch.bfloat16)])
def test_silu_in_gate_epilogue_equals_silu_in_down_producer(name, dtype):
    """act_seg0 / gate_activated (the gate tiles of the unpaired gate | up launch store round(silu(round(gate))), down's producer
    multiplies) against silu in down's producer (model.py:258-259 either way): the same bits in the down projection's slabs,
    the residual stream and the logits; the gate half of the hand-over buffer is silu of the other run's gate half."""
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine
    model = G.build_synthetic_model(name, DEV, dtype, seed=37, n_layer=2)
    ths = G.apply_sparsity(model, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True)
    prompt = torch.randint(0, model.config.vocab_size, (7,), device=DEV, dtype=torch.int, generator=torch.Generator(device=DEV).manual_seed(8))
    try:
        with torch.no_grad():
            model.max_seq_length = -1
            model.setup_caches(1, 32)
            model(prompt.view(1, -1), torch.arange(7, device=DEV))
            eng = DecodeEngine(model, ths, pair=False)
            assert eng.gate_act and not eng.pair
            tok = torch.tensor([[11]], device=DEV, dtype=torch.int)
            pos = torch.tensor([7], device=DEV, dtype=torch.int)
            inter = eng.inter

            def run(flag):
                eng.use_gate_act = flag
                eng._build(ths)
                assert eng.gate_act == flag
                logits = eng(tok, pos).clone()
                return logits, eng.s_down.clone(), eng.resid[0].clone(), eng.resid[1].clone(), eng.gu.clone()

            a, b = run(True), run(False)
            for i in range(4):
                assert torch.equal(a[i].view(torch.uint8), b[i].view(torch.uint8)), i
            assert torch.equal(a[4][inter:].view(torch.int16), b[4][inter:].view(torch.int16)), "up half"
            sg = torch.nn.functional.silu(b[4][:inter].float())  # last layer's gate, activation restated in fp32
            err = (a[4][:inter].float() - sg).abs()
            ulp = sg.abs().clamp_min(6e-5) * (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7)
            assert bool((err <= ulp).all()), float((err / ulp).max())
    finally:
        del model
        torch.cuda.empty_cache()
