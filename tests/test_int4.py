"""int4 group-quantised sparse GEMV (second half): the quantiser is pinned bit-exactly against the
reference's own group_quantize_tensor / group_dequantize_tensor (fixture F10, oracle/gen_golden.py:gen_int4); the HIP
kernel is checked against the float64 composition masked(x) @ dequant(W).T the fixture carries, and on real layer
widths against a float64 restatement."""
import numpy as np
import pytest
import torch

from helpers import load_kat

DEV = "cuda"


def _bf16(bits):
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)


@pytest.mark.parametrize("tag", ["g32", "g128"])
def test_int4_quantiser_bit_identical_to_reference(tag):
    from teal_amd.quantize import group_dequantize_tensor, group_quantize_tensor, pack_int4_colmajor, unpack_int4_colmajor
    k = load_kat("kat_int4.npz")
    N, Z, G, _, _ = (int(v) for v in k[f"{tag}_shape"])
    w = _bf16(k[f"{tag}_w"]).view(N, Z)
    q, sz = group_quantize_tensor(w, 4, G)
    assert np.array_equal(q.numpy().astype(np.uint8), k[f"{tag}_q"])
    assert np.array_equal(sz.view(torch.int16).numpy().view(np.uint16), k[f"{tag}_sz"]) and sz.shape == (Z // G, N, 2)
    wdq = group_dequantize_tensor(q, sz.float(), 4, G)
    assert np.array_equal(wdq[:16].float().numpy(), k[f"{tag}_wdq_rows16"])
    packed = pack_int4_colmajor(q)
    assert packed.shape[0] == Z // 2 and packed.dtype == torch.uint8 and torch.equal(unpack_int4_colmajor(packed, N), q)
    assert packed.stride(0) % 128 == 0, "pair-rows start on 128-byte lines: a tile's 128-byte segment must not straddle two"


def _truth(x, q, sz, G, cols):
    """float64: y[n] = sum over kept m of x[m] * ((q[n][m] - 8) * scale[m / G][n] + zero[m / G][n])"""
    N, Z = q.shape
    s = sz[:, :, 0].double().repeat_interleave(G, dim=0).T  # [N, Z]
    z = sz[:, :, 1].double().repeat_interleave(G, dim=0).T
    w = (q.double() - 8.0) * s + z
    y = torch.zeros(N, dtype=torch.float64)
    for c0, c1, tau in cols:
        if c1 > c0:
            xm = torch.where(x.float().abs() > torch.tensor(tau, dtype=torch.float32), x.double(), torch.zeros_like(x, dtype=torch.float64))
            y[c0:c1] = w[c0:c1] @ xm
    return y.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["g32", "g128"])
def test_int4_sparse_gemv_golden(oracle, tag):
    import teal_amd.kernels.sparse_gemv as K
    from teal_amd.quantize import pack_int4_colmajor
    from helpers import tolerance
    k = load_kat("kat_int4.npz")
    N, Z, G, N_q, N_kv = (int(v) for v in k[f"{tag}_shape"])
    q = torch.from_numpy(k[f"{tag}_q"].astype(np.int32))
    sz = _bf16(k[f"{tag}_sz"]).view(Z // G, N, 2)
    x = _bf16(k[f"{tag}_x"]).view(1, 1, Z)
    taus = [float(t) for t in k[f"{tag}_taus"]]
    packed = pack_int4_colmajor(q).to(DEV)
    for xdt, dcode in ((torch.bfloat16, 1), (torch.float16, 0)):
        xs = x.to(xdt)
        y = K.qkv_gemv_int4(xs.to(DEV), packed, sz.to(DEV), taus[0], taus[1], taus[2], 0, N_kv)
        got = y.view(-1).float().cpu().numpy().astype(np.float64)
        cols = [(0, N_q, taus[0]), (N_q, N_q + N_kv, taus[1]), (N_q + N_kv, N, taus[2])]
        exact = _truth(xs.view(-1), q, sz, G, cols)
        assert (np.abs(got - exact) <= tolerance(oracle, exact, dcode)).all(), float(np.abs(got - exact).max())
        if xdt == torch.bfloat16:
            # the reference's dequantised weight is rounded to bf16 (twice): our fp32 evaluation sits within that rounding
            ref = k[f"{tag}_y"]
            assert (np.abs(got - ref) <= 2 * tolerance(oracle, ref, dcode) + 4e-3 * np.abs(ref)).all()
            assert np.abs(got - exact).max() <= np.abs(ref - exact).max() + float(oracle.ulp16(exact, dcode).max())


@pytest.mark.gpu
@pytest.mark.parametrize("Z,N,G,kv", [(4096, 4096, 32, 0), (4096, 12288, 32, 4096), (11008, 4096, 64, 0), (4096, 11008, 128, 0), (4096, 6144, 32, 1024)])
def test_int4_sparse_gemv_layer_widths(oracle, Z, N, G, kv):
    """7B / 8B projection widths: 3-threshold qkv (MHA and GQA), wo, gate, down; split-K tickets; NaN propagation."""
    import teal_amd.kernels.sparse_gemv as K
    from teal_amd.quantize import group_quantize_tensor, pack_int4_colmajor
    from helpers import tolerance
    g = torch.Generator().manual_seed(Z + N + G)
    w = (torch.randn(N, Z, generator=g) * 0.03).to(torch.bfloat16)
    q, sz = group_quantize_tensor(w, 4, G)
    x = (torch.rand(Z, generator=g) * 4 - 2).to(torch.float16)
    tq, tk, tv = (1.0, 0.8, 1.2) if kv else (1.0, 1.0, 1.0)
    packed, szd = pack_int4_colmajor(q).to(DEV), sz.to(DEV)
    y = K.qkv_gemv_int4(x.view(1, 1, Z).to(DEV), packed, szd, tq, tk, tv, 0, kv)
    y2 = K.qkv_gemv_int4(x.view(1, 1, Z).to(DEV), packed, szd, tq, tk, tv, 0, kv)
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16)), "bit-reproducible"
    cols = [(0, N - 2 * kv, tq), (N - 2 * kv, N - kv, tk), (N - kv, N, tv)]
    exact = _truth(x, q, sz, G, cols)
    got = y.view(-1).float().cpu().numpy().astype(np.float64)
    assert (np.abs(got - exact) <= tolerance(oracle, exact, 0)).all(), float(np.abs(got - exact).max())
    xn = x.clone()
    xn[5] = float("nan")
    yn = K.splitk_sparse_gemv_int4(xn.view(1, 1, Z).to(DEV), packed, szd, 1.0)
    assert bool(torch.isnan(yn).all()), "a NaN activation poisons every output like the reference's 0 * NaN"


@pytest.mark.gpu
def test_int4_model_decode_matches_dequantised_dense():
    """quantize_model_int4 + monkeypatch: the module path on int4 blocks (every row kept) equals the same model with the
    dequantised weights in plain Linears; prefill runs the dense dequantised path."""
    from teal_amd.gpt_fast import generate as G
    from teal_amd.quantize import WeightOnlyInt4Linear, quantize_model_int4
    m = G.build_synthetic_model("tiny-test", DEV, torch.bfloat16, seed=3, std=0.05)
    ref = G.build_synthetic_model("tiny-test", DEV, torch.bfloat16, seed=3, std=0.05)
    quantize_model_int4(m, 32)
    for lm, lr in zip(m.layers, ref.layers):
        for sub, names in (("attention", ("wqkv", "wo")), ("feed_forward", ("w1", "w3", "w2"))):
            for n in names:
                qm = getattr(getattr(lm, sub), n)
                assert isinstance(qm, WeightOnlyInt4Linear)
                getattr(getattr(lr, sub), n).weight.data = qm.dequantized(torch.bfloat16)
    ths = G.apply_sparsity(m, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True)
    assert ths[0]["q"] == -1.0 and m.layers[0].attention.int4 and hasattr(m.layers[0].feed_forward, "gemv2")
    toks = torch.randint(0, 512, (6,), dtype=torch.int, generator=torch.Generator().manual_seed(4)).to(DEV)  # (seeded: the
    # tolerance below is a few bf16 ulps of the logits, and 30 random prompts spread over 1-2.3 ulps)
    for mod in (m, ref):
        mod.max_seq_length = -1
        mod.setup_caches(1, 32)
    with torch.no_grad():
        a = m(toks.view(1, -1), torch.arange(6, device=DEV))
        b = ref(toks.view(1, -1), torch.arange(6, device=DEV))
        assert torch.allclose(a.float(), b.float(), atol=2e-2, rtol=5e-2)
        t = torch.tensor([[7]], device=DEV, dtype=torch.int)
        b1 = ref(t, torch.tensor([6], device=DEV))
        for fused in (True, False):  # decode: the fused engine over the int4 blocks, then teal::sparse_gemv_int4 op by op
            m.fused_decode = fused
            a1 = m(t, torch.tensor([6], device=DEV)).clone()
            assert torch.allclose(a1.float(), b1.float(), atol=6e-2, rtol=5e-2), (fused, float((a1.float() - b1.float()).abs().max()))


def test_int4_model_quantiser_respects_the_kernel_shape_contract(tmp_path):
    """quantize_model_int4 only converts blocks whose five projections the sparse int4 kernel can take (128-column tiles,
    whole groups; teal_sparse_qkv_gemv_i4) — a model like stories15M (dim 288) stays in 16 bits instead of failing with
    TEAL_ERR_SHAPE at the first decode step (round-2 advice) — and convert_for_runtime_int4 builds the modules a saved
    int4 state dict loads into (the loader's *int4* branch, gpt-fast/generate.py:236-242)."""
    from teal_amd.gpt_fast import generate as G
    from teal_amd.quantize import (WeightOnlyInt4Linear, convert_for_runtime_int4, int4_kernel_supports, is_int4,
                                   quantize_model_int4)
    assert int4_kernel_supports(4096, 12288, 32, 4096) and int4_kernel_supports(4096, 6144, 128, 1024)
    assert not int4_kernel_supports(288, 864, 32, 288) and not int4_kernel_supports(4096, 4096, 48)
    assert not int4_kernel_supports(4096, 4160, 32, 32)
    small = quantize_model_int4(G.build_synthetic_model("stories15M", "cpu", torch.float16, seed=1), 32)
    assert not any(isinstance(m, WeightOnlyInt4Linear) for m in small.modules())
    m = quantize_model_int4(G.build_synthetic_model("tiny-test", "cpu", torch.float16, seed=1), 32)
    lay = m.layers[0]
    assert all(is_int4(l) for l in (lay.attention.wqkv, lay.attention.wo, lay.feed_forward.w1, lay.feed_forward.w3, lay.feed_forward.w2))
    assert not is_int4(m.output)  # lm_head stays in 16 bits
    path = tmp_path / "tiny-test" / "model_int4.g32.pth"
    path.parent.mkdir()
    torch.save(m.state_dict(), path)
    m2 = G.load_checkpoint_model(path, "cpu", torch.float16)
    assert is_int4(m2.layers[1].feed_forward.w2) and m2.layers[1].feed_forward.w2.groupsize == 32
    for (ka, va), (kb, vb) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    assert m2.output.weight.dtype == torch.float16 and m2.layers[0].attention.wqkv.scales_and_zeros.dtype == torch.bfloat16
    # dense forward of the loaded model = the quantised model's (dequantised matmul)
    m.setup_caches(1, 16)
    m2.setup_caches(1, 16)
    toks = torch.randint(0, 512, (5,), dtype=torch.int)
    with torch.no_grad():
        a = m(toks.view(1, -1), torch.arange(5))
        b = m2(toks.view(1, -1), torch.arange(5))
    assert torch.equal(a, b)


def _gap_tau(xabs: torch.Tensor, frac: float) -> float:
    """a threshold near the `frac` quantile of |x| sitting in the middle of the widest gap there: a producer that differs
    from the torch restatement by an ulp cannot flip a row across it"""
    v = torch.sort(xabs.float().cpu()).values
    k = int(frac * len(v))
    lo, hi = max(1, k - 24), min(len(v) - 1, k + 24)
    gaps = v[lo + 1: hi + 1] - v[lo: hi]
    g = int(torch.argmax(gaps)) + lo
    return float((v[g] + v[g + 1]) / 2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_int4_fused_producers_and_slabs(dtype):
    """teal_fused_gemv with weight_bits = 4 (include/teal_hip.h): each producer (RESID_NORM over interleaved slabs, SILU_MUL,
    ATTN_MERGE at 4 and 8 splits; each at a narrow output, where a wave's lane groups share its one or two units, and at a
    wide one, where each lane group takes a unit of its own) against the PLAIN int4 launch on the torch restatement of the activation vector, with the
    thresholds placed in gaps of |x|; resid_out bit-exact; the SLABS output summed in slice order equals the ROUNDED one
    bit for bit (same partials, same order)."""
    import ctypes
    from teal_amd import _lib, runtime
    from teal_amd.gpt_fast.engine import (TEAL_IN_ATTN_MERGE, TEAL_IN_PLAIN, TEAL_IN_RESID_NORM, TEAL_IN_SILU_MUL, TEAL_OUT_ROUNDED,
                                          TEAL_OUT_SLABS, GemvIn, _out)
    from teal_amd.quantize import group_quantize_tensor, pack_int4_colmajor
    L = _lib.load()
    runtime.init()
    code = runtime.dtype_code(dtype)
    g = torch.Generator().manual_seed(11)
    st = runtime.stream_ptr()

    def linear(Z, N, G):
        w = (torch.randn(N, Z, generator=g) * 0.03).to(torch.bfloat16)
        q, sz = group_quantize_tensor(w, 4, G)
        return pack_int4_colmajor(q).to(DEV), sz.to(DEV)

    def run(gin, Z, N, G, packed, sz, taus, mode, ws, slabs=None):
        """two threshold segments over one image; -> (y, nslabs)"""
        y = torch.zeros(N, device=DEV, dtype=dtype)
        h = N // 2
        segs = [(packed.data_ptr(), packed.stride(0), 0, h, taus[0], y.data_ptr(), sz.data_ptr(), (N, G)),
                (packed.data_ptr(), packed.stride(0), h, N - h, taus[1], y.data_ptr() + 2 * h, sz.data_ptr(), (N, G))]
        gout = _out(segs, mode, slabs)
        n = ctypes.c_int(0)
        rc = L.teal_fused_gemv(ctypes.byref(gin), ctypes.byref(gout), Z, code, ws.data_ptr(), ws.numel() * 4, ctypes.byref(n), st)
        _lib.check(rc, "teal_fused_gemv")
        torch.cuda.synchronize()
        return y, n.value

    def check(gin, x_ref, Z, N, G, tol):
        packed, sz = linear(Z, N, G)
        ws = runtime.new_workspace(Z, N)
        for frac in (0.0, 0.5):
            taus = (-1.0, -1.0) if frac == 0.0 else (_gap_tau(x_ref.abs(), 0.5), _gap_tau(x_ref.abs(), 0.4))
            y, n = run(gin, Z, N, G, packed, sz, taus, TEAL_OUT_ROUNDED, ws)
            plain = GemvIn(mode=TEAL_IN_PLAIN, x=x_ref.data_ptr())
            y_ref, n_ref = run(plain, Z, N, G, packed, sz, taus, TEAL_OUT_ROUNDED, ws)
            assert n == n_ref and (n > 1 or N > 8192), "split-K through the arrival tickets (narrow outputs)"
            assert torch.allclose(y.float(), y_ref.float(), atol=tol, rtol=tol), (frac, float((y.float() - y_ref.float()).abs().max()))
            slabs = torch.full((N, (n + 3) & ~3), float("nan"), device=DEV, dtype=torch.float32)
            _, n2 = run(gin, Z, N, G, packed, sz, taus, TEAL_OUT_SLABS, ws, slabs)
            assert n2 == n
            t = torch.zeros(N, device=DEV, dtype=torch.float32)
            for j in range(n):
                t = t + slabs[:, j]
            assert torch.equal(t.to(dtype).view(torch.int16), y.view(torch.int16)), "slabs summed in slice order == rounded output"

    tol = 6e-3 if dtype == torch.float16 else 4e-2
    # RESID_NORM: h = resid + round(sum of 3 slabs); x = round(round(h * rstd) * w)
    Z, N, G = 4096, 512, 32
    resid = torch.randn(Z, generator=g).to(dtype).to(DEV)
    sl = torch.zeros(Z, 4, dtype=torch.float32)
    sl[:, :3] = torch.randn(Z, 3, generator=g) * 0.3
    sl = sl.to(DEV)
    nw = (1.0 + 0.1 * torch.randn(Z, generator=g)).to(dtype).to(DEV)
    rout = torch.zeros(Z, device=DEV, dtype=dtype)
    ysum = ((sl[:, 0] + sl[:, 1]) + sl[:, 2]).to(dtype)
    h = (resid.float() + ysum.float()).to(dtype)
    hf = h.float()
    x_ref = ((hf * torch.rsqrt(hf.pow(2).mean() + 1e-5)).to(dtype).float() * nw.float()).to(dtype)
    gin = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=resid.data_ptr(), slabs=sl.data_ptr(), nslabs=3, slabs_interleaved=1,
                 norm_weight=nw.data_ptr(), eps=1e-5, resid_out=rout.data_ptr())
    check(gin, x_ref, Z, N, G, tol)
    assert torch.equal(rout.view(torch.int16), h.view(torch.int16)), "updated residual, bit-exact"
    check(gin, x_ref, Z, 12288, G, tol)  # 96 tiles x 2 slices: four units per wave, one per 16-lane group
    # other widths of the producer: 5120 (8 elements per thread, the last block ragged; 8 slabs = two 16-byte loads per element)
    # and 9216 (16 per thread, in blocks of four; one slab)
    for Zw, ns in ((5120, 8), (9216, 1)):
        r2 = torch.randn(Zw, generator=g).to(dtype).to(DEV)
        s2 = torch.zeros(Zw, (ns + 3) & ~3, dtype=torch.float32)
        s2[:, :ns] = torch.randn(Zw, ns, generator=g) * 0.2
        s2 = s2.to(DEV)
        w2 = (1.0 + 0.1 * torch.randn(Zw, generator=g)).to(dtype).to(DEV)
        ro2 = torch.zeros(Zw, device=DEV, dtype=dtype)
        acc = torch.zeros(Zw, device=DEV, dtype=torch.float32)
        for j in range(ns):
            acc = acc + s2[:, j]
        h2 = (r2.float() + acc.to(dtype).float()).to(dtype)
        x2 = ((h2.float() * torch.rsqrt(h2.float().pow(2).mean() + 1e-5)).to(dtype).float() * w2.float()).to(dtype)
        gin2 = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=r2.data_ptr(), slabs=s2.data_ptr(), nslabs=ns, slabs_interleaved=1,
                      norm_weight=w2.data_ptr(), eps=1e-5, resid_out=ro2.data_ptr())
        check(gin2, x2, Zw, 256, 64, tol)
        assert torch.equal(ro2.view(torch.int16), h2.view(torch.int16)), Zw
    # embedding lookup form (layer 0): row_index, no slabs
    table = torch.randn(7, Z, generator=g).to(dtype).to(DEV)
    idx = torch.tensor([5], device=DEV, dtype=torch.int32)
    hf = table[5].float()
    x_ref = ((hf * torch.rsqrt(hf.pow(2).mean() + 1e-5)).to(dtype).float() * nw.float()).to(dtype)
    gin = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=table.data_ptr(), row_index=idx.data_ptr(), nslabs=0, slabs_interleaved=1,
                 norm_weight=nw.data_ptr(), eps=1e-5, resid_out=rout.data_ptr())
    check(gin, x_ref, Z, N, G, tol)
    assert torch.equal(rout, table[5])
    # SILU_MUL over a width that is no multiple of 1024 (several blocks of loads per thread, a ragged last one)
    Z, N, G = 5504, 256, 64
    gu = torch.randn(2 * Z, generator=g).to(dtype).to(DEV)
    x_ref = (torch.nn.functional.silu(gu[:Z].float()).to(dtype).float() * gu[Z:].float()).to(dtype)
    check(GemvIn(mode=TEAL_IN_SILU_MUL, x=gu.data_ptr()), x_ref, Z, N, G, tol)
    check(GemvIn(mode=TEAL_IN_SILU_MUL, x=gu.data_ptr()), x_ref, Z, 11008, G, tol)  # 86 tiles x 2 slices: two passes, the second ragged
    # ATTN_MERGE: partials {max, sum, o[hd]} per (head, split)
    for ns, hd in ((4, 128), (8, 64)):
        Z, N, G = 2048, 512, 128
        nh = Z // hd
        p = torch.zeros(nh, ns, hd + 2)
        p[:, :, 0] = torch.randn(nh, ns, generator=g)
        p[:, :, 1] = torch.rand(nh, ns, generator=g) * 1.5 + 0.5
        p[:, :, 2:] = torch.randn(nh, ns, hd, generator=g) * p[:, :, 1:2]
        p[1, 2, 1] = 0.0  # an empty split contributes nothing
        p = p.to(DEV)
        m, l, o = p[:, :, 0], p[:, :, 1], p[:, :, 2:]
        f = torch.where(l > 0, torch.exp(m - m.max(dim=1, keepdim=True).values), torch.zeros_like(m))
        x_ref = ((o * (f / (l * f).sum(1, keepdim=True))[:, :, None]).sum(1)).reshape(-1).to(dtype)
        check(GemvIn(mode=TEAL_IN_ATTN_MERGE, x=p.data_ptr(), att_head_dim=hd, att_nsplit=ns), x_ref, Z, N, G, tol)
        check(GemvIn(mode=TEAL_IN_ATTN_MERGE, x=p.data_ptr(), att_head_dim=hd, att_nsplit=ns), x_ref, Z, 16640, G, tol)  # 130 tiles, no split
    # contract: planar / more than 8 input slabs and the paired output have no int4 form
    packed, sz = linear(4096, 256, 32)
    ws = runtime.new_workspace(4096, 256)
    y = torch.zeros(256, device=DEV, dtype=dtype)
    bad = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=resid.data_ptr(), slabs=sl.data_ptr(), nslabs=3, slabs_interleaved=0,
                 norm_weight=nw.data_ptr(), eps=1e-5, resid_out=rout.data_ptr())
    gout = _out([(packed.data_ptr(), packed.stride(0), 0, 256, 0.5, y.data_ptr(), sz.data_ptr(), (256, 32))], TEAL_OUT_ROUNDED)
    assert L.teal_fused_gemv(ctypes.byref(bad), ctypes.byref(gout), 4096, code, ws.data_ptr(), ws.numel() * 4, None, st) != 0
    gout.groupsize = 48
    ok_in = GemvIn(mode=TEAL_IN_PLAIN, x=resid.data_ptr())
    assert L.teal_fused_gemv(ctypes.byref(ok_in), ctypes.byref(gout), 4096, code, ws.data_ptr(), ws.numel() * 4, None, st) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,sparsity", [(torch.float16, 0.0), (torch.bfloat16, 0.0), (torch.float16, 0.5)])
def test_int4_engine_matches_int4_module_path(dtype, sparsity):
    """DecodeEngine over int4 group-quantised blocks (5 launches per layer through teal_fused_gemv with weight_bits = 4, one
    hipGraph replay per token) against the op-by-op module path on the same int4 blocks: equal to rounding with every row
    kept, cosine > 0.98 at 50 % (near-tau activations may flip); the captured step replays bit-identically; a single-token
    call of the patched model uses it."""
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine, pick_engine
    from teal_amd.quantize import quantize_model_int4
    ref = quantize_model_int4(G.build_synthetic_model("tiny-test", DEV, dtype, seed=3, std=0.05), 32)
    ref.fused_decode = False  # op-by-op module path
    eng_m = quantize_model_int4(G.build_synthetic_model("tiny-test", DEV, dtype, seed=3, std=0.05), 32)
    ths = G.apply_sparsity(ref, sparsity=sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
    G.apply_sparsity(eng_m, sparsity=sparsity, hist_path=None, greedy_lookup=None, synthetic=True, decode_calibration=False)
    prompt = torch.tensor([5, 17, 301, 44, 9], device=DEV, dtype=torch.int)
    with torch.no_grad():
        for m in (ref, eng_m):
            m.max_seq_length = -1
            m.setup_caches(1, 64)
            m(prompt.view(1, -1), torch.arange(5, device=DEV))
        assert pick_engine(eng_m)[0] is DecodeEngine
        eng = DecodeEngine(eng_m, ths)
        assert eng.int4 and not eng.pair and eng.att_fused_merge
        for step, tok_id in enumerate((7, 100, 3)):
            tok = torch.tensor([[tok_id]], device=DEV, dtype=torch.int)
            pos = torch.tensor([5 + step], device=DEV, dtype=torch.int)
            a, b = ref(tok, pos).float().view(-1), eng(tok, pos).float().view(-1)
            if sparsity == 0.0:
                tol = 8e-3 if dtype == torch.float16 else 6e-2
                assert torch.allclose(a, b, atol=tol, rtol=tol), (step, float((a - b).abs().max()))
            else:
                assert torch.nn.functional.cosine_similarity(a, b, dim=0) > 0.98
        for la, lb in zip(ref.layers, eng_m.layers):  # both appended the same K/V rows
            if sparsity == 0.0:
                tolk = 2e-2 if dtype == torch.float16 else 8e-2
                assert torch.allclose(la.attention.kv_cache.k_cache.float(), lb.attention.kv_cache.k_cache.float(), atol=tolk, rtol=2e-2)
        # the device-resident loop: one hipGraph replay per token, same tokens as eager launches
        first = torch.tensor([11], device=DEV, dtype=torch.int)
        eng.manual_seed(5)
        t_graph = eng.decode_n(first, 8, 6, temperature=0.8, top_k=50, use_graph=True)
        eng.manual_seed(5)
        t_eager = eng.decode_n(first, 8, 6, temperature=0.8, top_k=50, use_graph=False)
        assert torch.equal(t_graph, t_eager)
        kept = eng.kept_fractions(torch.tensor([[7]], device=DEV, dtype=torch.int), torch.tensor([14], device=DEV, dtype=torch.int))
        assert set(kept) == set(eng.SITE) and all(0.0 <= v <= 1.0 for v in kept.values())
        # a single-token call of the patched model takes the same engine
        eng_m(torch.tensor([[7]], device=DEV, dtype=torch.int), torch.tensor([15], device=DEV))
        assert isinstance(eng_m._eng, DecodeEngine) and eng_m._eng.int4


def test_quantize_cli_writes_the_checkpoints_the_loader_branches_on(tmp_path):
    """teal_amd.quantize.quantize (the role of gpt-fast/quantize.py:528-600): model.pth -> model_int8.pth and
    model_int4.g32.pth next to it, under the names load_checkpoint_model keys on; the state dicts load into the
    convert_for_runtime_* shells and reproduce the quantiser's own tensors (CPU only)."""
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.model import Transformer
    from teal_amd.quantize import (convert_for_runtime_int4, convert_for_runtime_int8, quantize, quantize_model_int4,
                                   quantize_model_int8)
    ck = tmp_path / "tiny-test"
    ck.mkdir()
    torch.save(G.build_synthetic_model("tiny-test", "cpu", torch.bfloat16, seed=9, std=0.05).state_dict(), ck / "model.pth")
    p8 = quantize(ck / "model.pth", "int8", label="_")
    p4 = quantize(ck / "model.pth", "int4", groupsize=32, label="_")
    assert p8.name == "model_int8.pth" and p4.name == "model_int4.g32.pth"
    with pytest.raises(ValueError, match="int4-gptq"):
        quantize(ck / "model.pth", "int4-gptq")
    base = lambda: G.build_synthetic_model("tiny-test", "cpu", torch.bfloat16, seed=9, std=0.05)  # noqa: E731
    for path, conv, ref in ((p8, lambda m: convert_for_runtime_int8(m, torch.bfloat16), quantize_model_int8(base())),
                            (p4, lambda m: convert_for_runtime_int4(m, 32), quantize_model_int4(base(), 32))):
        with torch.device("meta"):
            m = Transformer.from_name("tiny-test")
        conv(m)
        m.load_state_dict(torch.load(str(path), weights_only=True), assign=True)
        want = ref.state_dict()
        got = m.state_dict()
        assert set(got) == set(want)
        for k in want:
            assert torch.equal(got[k].cpu(), want[k].cpu()), k
