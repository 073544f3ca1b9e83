"""Decode harness: CPU checks of the model/generate plumbing (dense path only — the sparse ops are
GPU-only) and GPU checks of the monkeypatched model under hipGraph capture."""
import os

import numpy as np
import pytest
import torch

from teal_amd.gpt_fast import generate as G
from teal_amd.gpt_fast.model import ModelArgs, Transformer, apply_rotary_emb, precompute_freqs_cis


def tiny(device, dtype=torch.float32):
    return G.build_synthetic_model("tiny-test", device, dtype, seed=3, std=0.05)


def test_model_args_fuzzy_match_and_shapes():
    a = ModelArgs.from_name("Llama-2-7b-chat-hf")
    assert (a.n_layer, a.dim, a.intermediate_size, a.n_local_heads) == (32, 4096, 11008, 32)
    b = ModelArgs.from_name("Meta-Llama-3-8B")
    assert (b.intermediate_size, b.n_local_heads, b.vocab_size, b.rope_base) == (14336, 8, 128256, 500000)
    c = ModelArgs.from_name("llama-2-70B")
    assert (c.n_layer, c.dim, c.intermediate_size, c.n_local_heads) == (80, 8192, 28672, 8)
    assert ModelArgs.from_name("Mistral-7B-v0.1").n_local_heads == 8  # longest match beats "7B"


def test_rope_is_a_rotation_and_position_zero_is_identity():
    fc = precompute_freqs_cis(16, 8, 10000, torch.float32)
    x = torch.randn(1, 16, 2, 8)
    y = apply_rotary_emb(x, fc)
    assert torch.allclose(y[:, 0], x[:, 0], atol=1e-6)
    assert torch.allclose(y.norm(dim=-1), x.norm(dim=-1), atol=1e-5)


def test_decode_matches_prefill_cpu_dense():
    """KV-cache decode of token t equals the t-th position of a full prefill."""
    m = tiny("cpu")
    m.setup_caches(1, 16)
    toks = torch.randint(0, 512, (8,), dtype=torch.int)
    with torch.no_grad():
        full = m(toks.view(1, -1), torch.arange(8))
        m2 = tiny("cpu")
        m2.setup_caches(1, 16)
        m2(toks[:5].view(1, -1), torch.arange(5))
        outs = [m2(toks[i].view(1, 1), torch.tensor([i])) for i in range(5, 8)]
    for j, o in enumerate(outs):
        assert torch.allclose(o[0, 0], full[0, 5 + j], atol=2e-4, rtol=1e-3)


def test_generate_cpu_dense_is_deterministic_under_seed():
    m = tiny("cpu")
    prompt = torch.randint(0, 512, (6,), dtype=torch.int)
    outs = []
    for _ in range(2):
        torch.manual_seed(1234)
        dec = G.GraphedDecoder(m, False, 0.8, 50)
        outs.append(G.generate(m, prompt, 12, dec, temperature=0.8, top_k=50))
        m.max_seq_length = -1
    assert torch.equal(outs[0], outs[1]) and outs[0].numel() == 18 and torch.equal(outs[0][:6], prompt)


def test_calibration_hits_target_kept_fraction_cpu():
    m = tiny("cpu")
    L = len(m.layers)
    sp = {p: [0.5] * L for p in G.PROJS}
    ths = G.calibrate_thresholds(m, sp, n_tokens=64)
    assert len(ths) == L and all(t["q"] == t["k"] == t["v"] > 0 and t["gate"] == t["up"] > 0 for t in ths)
    assert G.calibrate_thresholds(m, {p: [0.0] * L for p in G.PROJS})[0]["down"] == -1.0
    # on fresh tokens the kept fraction of the MLP input is ~50 %
    acts = []
    h = m.layers[1].feed_forward.register_forward_pre_hook(lambda mod, a: acts.append(a[0].flatten()))
    m.setup_caches(1, 64)
    with torch.no_grad():
        m(torch.randint(0, 512, (64,), dtype=torch.int).view(1, -1), torch.arange(64))
    h.remove()
    kept = (torch.cat(acts).abs() > ths[1]["gate"]).float().mean()
    assert 0.4 < float(kept) < 0.6


def test_cli_flags_match_reference_surface():
    p = G.build_parser()
    a = p.parse_args(["--hist_path", "H", "--sparsity", "0.5", "--compile", "--num_samples", "2", "--max_new_tokens", "10"])
    assert a.hist_path == "H" and a.sparsity == 0.5 and a.compile and a.max_new_tokens == 10
    d = p.parse_args([])
    assert d.max_new_tokens == 200 and d.num_samples == 5 and d.top_k == 200 and d.temperature == 0.8 and d.sparsity == 0.0
    # every flag of the reference's parser (gpt-fast/generate.py:532-548) is accepted with its default
    assert d.prompt == "Hello, my name is" and not d.interactive and d.speculate_k == 5 and d.draft_checkpoint_path is None
    assert not d.compile_prefill and d.profile is None and str(d.checkpoint_path).endswith("Llama-2-7b-chat-hf/model.pth")
    full = p.parse_args(["--prompt", "x", "--interactive", "--num_samples", "1", "--max_new_tokens", "3", "--top_k", "5", "--temperature", "0.5",
                         "--checkpoint_path", "a/model.pth", "--compile", "--compile_prefill", "--profile", "t", "--speculate_k", "4",
                         "--draft_checkpoint_path", "d/model.pth", "--device", "cuda", "--hist_path", "H", "--sparsity", "0.4"])
    assert full.interactive and full.speculate_k == 4
    with pytest.raises(SystemExit, match="speculative decoding"):  # named, not silently ignored
        G.main(full)
    with pytest.raises(SystemExit, match="tokenizer"):
        G.main(p.parse_args(["--synthetic", "tiny-test", "--interactive", "--device", "cuda"]))


def test_decode_engine_supports_names_ineligible_models():
    """DecodeEngine.supports(): the predicate Transformer.forward and the synthetic calibration consult before building
    the fused step; an ineligible model keeps the op-by-op module path instead of raising inside forward."""
    from teal_amd.gpt_fast.engine import DecodeEngine
    m = G.build_synthetic_model("stories15M", "cpu", torch.float16, seed=1)  # dim 288, 6 heads: head_dim 48
    m.setup_caches(1, 16)
    assert "head_dim 48" in DecodeEngine.supports(m)
    t = tiny("cpu", torch.float16)  # head_dim 64, dim 256
    assert "setup_caches" in DecodeEngine.supports(t)
    t.setup_caches(2, 16)
    assert "max_batch_size == 1" in DecodeEngine.supports(t)
    t.max_seq_length = -1
    t.max_batch_size = -1
    t.setup_caches(1, 16)
    assert DecodeEngine.supports(t) == "model is not on a HIP device"  # everything else is fine
    w2 = t.layers[1].feed_forward.w2
    w2.weight = torch.nn.Parameter(w2.weight.data.to(torch.int8), requires_grad=False)
    assert "mixed int8" in DecodeEngine.supports(t)
    with pytest.raises(ValueError, match="cannot run this model"):
        DecodeEngine(t, [])


# --------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_ineligible_model_keeps_the_module_path():
    """stories15M (head_dim 48): patching and single-token calls must work through the op-by-op path, as they did before
    the fused engine existed (round-2 advice: the engine was built unconditionally and raised TEAL_ERR_SHAPE)."""
    dev = "cuda"
    m = G.build_synthetic_model("stories15M", dev, torch.float16, seed=5, std=0.05)
    ths = G.apply_sparsity(m, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True)  # no engine: no decode refinement
    assert len(ths) == len(m.layers) and ths[0]["q"] > 0
    m.setup_caches(1, 32)
    V = m.config.vocab_size
    toks = torch.randint(0, V, (6,), device=dev, dtype=torch.int)
    with torch.no_grad():
        m(toks.view(1, -1), torch.arange(6, device=dev))
        t = torch.tensor([[7]], device=dev, dtype=torch.int)
        a = m(t, torch.tensor([6], device=dev))          # fused_decode is on, the engine declines: module path
        assert getattr(m, "_eng", None) is None and "head_dim 48" in m._eng_why
        m.fused_decode = False
        b = m(t, torch.tensor([6], device=dev))
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_fused_forward_returns_a_fresh_tensor():
    dev = "cuda"
    m = tiny(dev, torch.float16)
    G.apply_sparsity(m, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True)
    m.setup_caches(1, 32)
    with torch.no_grad():
        a = m(torch.tensor([[3]], device=dev, dtype=torch.int), torch.tensor([0], device=dev))
        keep = a.clone()
        b = m(torch.tensor([[9]], device=dev, dtype=torch.int), torch.tensor([1], device=dev))
    assert m._eng is not None and a.data_ptr() != b.data_ptr() and torch.equal(a, keep)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_monkeypatched_model_matches_dense_when_everything_is_kept(fused):
    """fused: a single-token call on the patched model runs the fused HIP decode step (Transformer.fused_decode);
    otherwise the op-by-op module path (torch.ops.teal.* + eager glue)."""
    dev = "cuda"
    m = tiny(dev, torch.float16)
    m.fused_decode = fused
    ref = tiny(dev, torch.float16)
    G.apply_sparsity(m, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True)  # tau = -1: dense on the HIP kernels
    toks = torch.randint(0, 512, (6,), device=dev, dtype=torch.int)
    for mod in (m, ref):
        mod.setup_caches(1, 32)
    with torch.no_grad():
        a = m(toks.view(1, -1), torch.arange(6, device=dev))       # prefill: ops fall back to matmul
        b = ref(toks.view(1, -1), torch.arange(6, device=dev))
        assert torch.allclose(a, b, atol=2e-3, rtol=2e-2)
        t = torch.tensor([[7]], device=dev, dtype=torch.int)
        a1 = m(t, torch.tensor([6], device=dev))                    # decode: HIP sparse GEMV path
        b1 = ref(t, torch.tensor([6], device=dev))
        assert torch.allclose(a1.float(), b1.float(), atol=4e-3, rtol=3e-2)
        assert (getattr(m, "_eng", None) is not None) == fused
    # the reference's attribute bundle is intact on every block either way (gpt-fast/generate.py:266-331)
    for layer in m.layers:
        at, ff = layer.attention, layer.feed_forward
        assert all(hasattr(at, n) for n in ("gemv1", "gemv2", "gemv1_kernel", "gemv2_kernel", "thresh_q", "thresh_k", "thresh_v", "thresh_o", "sparsity_bin", "old_forward"))
        assert all(hasattr(ff, n) for n in ("gemv1", "gemv2", "gemv1_kernel", "gemv2_kernel", "thresh_gate", "thresh_up", "thresh_down", "sparsity_bin", "old_forward"))
        assert at.wqkv.weight.stride(0) == 1 and ff.w2.weight.stride(0) == 1  # column-major weights (generate.py:296-317)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_sparse_decode_graph_replay_equals_eager(fused):
    """the monkeypatched model's decode step, replayed from a hipGraph, is bit-identical to eager."""
    dev = "cuda"
    m = tiny(dev, torch.float16)
    m.fused_decode = fused
    G.apply_sparsity(m, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True)
    m.setup_caches(1, 32)
    prompt = torch.randint(0, 512, (6,), device=dev, dtype=torch.int)
    with torch.no_grad():
        m(prompt.view(1, -1), torch.arange(6, device=dev))
        tok = torch.tensor([[9]], device=dev, dtype=torch.int)
        pos = torch.tensor([6], device=dev, dtype=torch.int)
        eager = m(tok, pos).clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(tok, pos)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(tok, pos)
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), eager.view(torch.int16))


@pytest.mark.gpu
def test_sparse_thresholds_reduce_rows_read():
    """the achieved kept fraction of ALL SEVEN projections, measured on the DECODE activations of the fused engine, is
    near the calibrated target (and 1.0 with sparsity 0)."""
    from teal_amd.gpt_fast.engine import DecodeEngine
    dev = "cuda"
    for sparsity in (0.5, 0.0):
        m = tiny(dev, torch.float16)
        ths = G.apply_sparsity(m, sparsity=sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
        m.max_seq_length = -1
        m.setup_caches(1, 128)
        prompt = torch.randint(0, m.config.vocab_size, (24,), device=dev, dtype=torch.int)
        with torch.no_grad():
            m(prompt.view(1, -1), torch.arange(24, device=dev))
            eng = DecodeEngine(m, ths)
            # synthetic thresholds are taken on a decode of ~100 positions after a 24-token prompt
            # (generate.refine_thresholds_on_decode): measure over the same kind of range, 5 positions
            kf = eng.mean_kept_fractions(prompt[-1:].clone(), 24, 90, 5)
            one = eng.kept_fractions(torch.tensor([[5]], device=dev, dtype=torch.int), torch.tensor([60], device=dev, dtype=torch.int))
        assert set(kf) == set(one) == {"q", "k", "v", "o", "gate", "up", "down"}
        if sparsity == 0.0:
            assert all(v == 1.0 for v in kf.values()) and all(v == 1.0 for v in one.values()), kf
        else:
            # dim-256 toy model: per projection within 0.1 of the target, model-wide mean within 0.04
            assert all(abs(v - 0.5) < 0.1 for v in kf.values()), kf
            assert abs(sum(kf.values()) / 7 - 0.5) < 0.04, kf


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--engine"], [], ["--no_engine"], ["--no_engine", "--no_fused_decode"], ["--precision", "bf16"],
                                   ["--engine", "--sparsity", "0.0"], ["--compile_prefill"]])
def test_generate_main_synthetic_cli(extra):
    """the reference-shaped CLI end to end on a tiny synthetic model: load -> monkeypatch -> capture ->
    timed samples, through the fused engine and through the module path."""
    args = G.build_parser().parse_args(["--synthetic", "tiny-test", "--sparsity", "0.5", "--compile", "--num_samples", "2",
                                        "--max_new_tokens", "16", "--top_k", "50", "--report_kept"] + extra)
    res = G.main(args)
    assert res["decoder"] == ("GraphedDecoder" if "--no_engine" in extra else "EngineDecoder")  # --compile implies the fused engine
    assert len(res["tokens_per_sec"]) == 2 and all(t > 0 for t in res["tokens_per_sec"])
    assert res["thresholds"] is not None and len(res["thresholds"]) == 2


@pytest.mark.gpu
def test_compile_prefill_generates_the_same_tokens():
    """--compile captures the prompt pass too (the hand-fused HIP pass for <= 16 tokens, the patched modules under a hipGraph
    otherwise): the prefill graph is captured before the engine exists; the engine then re-lays the weights out (freeing the
    storage a stale graph would still point at).  Every timed sample must produce the tokens of the run WITHOUT a prefill graph
    (--eager_prefill: the op-by-op pass), across several samples (allocator reuse in between) — for the default pass and for
    the graphed module pass (--module_prefill), each against the eager baseline."""
    base = ["--synthetic", "tiny-test", "--sparsity", "0.5", "--compile", "--num_samples", "3", "--max_new_tokens", "24", "--top_k", "1"]
    eager = G.main(G.build_parser().parse_args(base + ["--eager_prefill"]))
    assert eager["prefill"] is None and len(eager["sequences"]) == 3
    junk = [torch.randn(1 << 16, device="cuda") for _ in range(16)]  # churn the caching allocator between the runs
    module = G.main(G.build_parser().parse_args(base + ["--module_prefill"]))
    assert module["prefill"] == "GraphedPrefill", module["prefill"]
    assert module["sequences"] == eager["sequences"]  # the same torch ops, replayed from a graph: the same tokens
    junk2 = [torch.randn(1 << 15, device="cuda") for _ in range(16)]
    default = G.main(G.build_parser().parse_args(base))
    assert str(default["prefill"]).startswith("FusedPrefill"), default["prefill"]
    del junk, junk2
    if default["prefill"].endswith(":hip"):
        # the HIP pass accumulates in another order than torch's matmuls: its last-position logits agree to a few output ulps
        # (tests/test_prefill.py), so under top_k = 1 the FIRST token may differ only on a near-tie; every sample of the run must
        # at least agree with the run's own first sample (same prompt, same graphs, allocator reuse in between)
        assert all(sq == default["sequences"][0] for sq in default["sequences"])
        agree = sum(a == b for a, b in zip(default["sequences"][0], eager["sequences"][0]))
        assert agree >= 7, (agree, default["sequences"][0], eager["sequences"][0])  # the 6 prompt tokens + at least the first new one
    else:
        assert default["sequences"] == eager["sequences"]


@pytest.mark.gpu
def test_generate_greedy_fixture_table():
    """block-wise greedy sparsities from the reference-derived table (tests/golden/greedy_llama2_7b.json):
    three distinct q/k/v thresholds and gate != up reach the kernels."""
    import os
    from helpers import GOLDEN
    m = G.build_synthetic_model("tiny-test", "cuda", torch.float16, seed=3, std=0.05)
    ths = G.apply_sparsity(m, sparsity=0.5, hist_path=None, greedy_lookup=os.path.join(GOLDEN, "greedy_llama2_7b.json"),
                           synthetic=True)
    assert len({ths[0]["q"], ths[0]["k"], ths[0]["v"]}) == 3 and ths[0]["gate"] != ths[0]["up"]
    at = m.layers[0].attention
    assert (at.thresh_q, at.thresh_k, at.thresh_v) == (ths[0]["q"], ths[0]["k"], ths[0]["v"])


@pytest.mark.gpu
def test_graphed_prefill_equals_eager_prefill():
    """--compile_prefill: the captured prompt pass returns the same logits and fills the same KV cache."""
    m = G.build_synthetic_model("tiny-test", "cuda", torch.float16, seed=3, std=0.05)
    G.apply_sparsity(m, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True)
    m.max_seq_length = -1
    m.setup_caches(1, 64)
    pre = G.GraphedPrefill(m)
    for seed, T in ((1, 6), (2, 6), (3, 11)):
        prompt = torch.randint(0, 512, (T,), device="cuda", dtype=torch.int, generator=torch.Generator(device="cuda").manual_seed(seed))
        with torch.no_grad():
            want = m(prompt.view(1, -1), torch.arange(T, device="cuda")).clone()
            kc = m.layers[0].attention.kv_cache.k_cache.clone()
            m.layers[0].attention.kv_cache.k_cache.zero_()
            got = pre(prompt)
            junk = [torch.randn(4096, device="cuda") for _ in range(8)]  # allocator churn: the graph's inputs must survive it
            torch.cuda.synchronize()
        assert torch.equal(got, want) and len(junk) == 8
        assert torch.equal(m.layers[0].attention.kv_cache.k_cache[:, :, :T], kc[:, :, :T])
    assert len(pre.graphs) == 2  # one graph per prompt length


@pytest.mark.gpu
def test_fused_decode_follows_replaced_weights_and_caches():
    """the fused step is rebuilt when the storage it points at goes away: weights replaced wholesale, KV caches
    re-allocated at the same size — the logits follow, and equal the op-by-op path on the same model."""
    dev = "cuda"
    m = tiny(dev, torch.float16)
    G.apply_sparsity(m, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True)
    toks = torch.randint(0, 512, (6,), device=dev, dtype=torch.int)
    t, p = torch.tensor([[7]], device=dev, dtype=torch.int), torch.tensor([6], device=dev, dtype=torch.int)

    def step():
        with torch.no_grad():
            m(toks.view(1, -1), torch.arange(6, device=dev))
            a = m(t, p).clone()
            m.fused_decode = False
            b = m(t, p).clone()
            m.fused_decode = True
        assert torch.allclose(a.float(), b.float(), atol=4e-3, rtol=3e-2)
        return a

    m.setup_caches(1, 32)
    a0 = step()
    eng0 = m._eng
    w = m.layers[0].feed_forward.w2
    w.weight = torch.nn.Parameter((w.weight.detach() * 0.5).contiguous(), requires_grad=False)  # new storage, row-major again
    from teal_amd.monkeypatch import to_column_major
    to_column_major(w)
    a1 = step()  # (the allocator may hand the new image the old block: then the same engine is still right)
    assert not torch.equal(a0, a1) and eng0 is not None
    m.max_seq_length = -1
    m.setup_caches(1, 32)  # same size, new cache tensors
    a2 = step()
    assert torch.allclose(a1.float(), a2.float(), atol=1e-3)


@pytest.mark.gpu
def test_bench_replicas_under_torch_distributed_run():
    """`bench.py --gpus 2` the way the driver launches it (torch.distributed.run, one rank per replica): every rank decodes its
    own stream, the only collectives are the timing barrier and the max over ranks, rank 0 prints one JSON line whose value is the
    aggregate.  On a one-GPU box the two replicas share the device (gloo; RCCL refuses two ranks on a GPU) — the flow is the same."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--n_layer", "2",
           "--no-cpu-baseline", "--no-context-sweep"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 10 / (d["ms_per_step"] * 10 / 1e3)) < 1e-6 * d["value"]
    # the dominant launch's roofline is a per-GPU property: rank 0 measures it after the ranks have left the group; the CPU
    # baseline is an N = 1 item
    assert d["cpu_baseline"] is None and "replicas x2" in d["config"]["parallelism"]
    assert d["roofline"]["bound"] == "hbm" and 0.05 < d["roofline"]["frac"] < 1.0 and d["roofline"]["peak"] == 8000.0
