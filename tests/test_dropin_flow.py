"""GPU: the reference harness's literal flow, end to end, on the HIP path (round-2 verdict item 4).

    generate.py --checkpoint_path <dir>/<model-name>/model.pth --hist_path <teal>/histograms --sparsity s --compile
    (gpt-fast/generate.py:225-254 _load_model incl. the int8 branch, :266-331 monkeypatch_layer, :328-331 the loop over
    the blocks, :420 the compiled decode step)

on a checkpoint and a histogram tree written by this repo's own producers (teal_amd/calibrate.py, the restatement of
teal/grab_acts.py) — there is no network for real checkpoints.  Asserted: every installed thresh_* equals
Distribution(path, h).icdf(0.5 + 0.5 s); the device-resident engine loop, the reference-harness loop over the patched
model and the op-by-op module path generate the same tokens; an *int8* checkpoint name takes the int8 branch.  A second
test patches a Llama-2-7B-width block with the committed Llama-2-7B histograms (fixture F6) and checks the thresholds
against fixture F1 and the patched forwards against a masked-matmul restatement.
"""
import argparse
import json
import os
from pathlib import Path

import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda"
HIST = os.path.join(GOLDEN, "hist", "Llama-2-7B")


def _tokenizer_model(dirpath: Path, vocab: int = 512) -> None:
    """a tiny sentencepiece model with exactly the model's 512 ids (the harness reads <checkpoint dir>/tokenizer.model and
    decodes what it generates, like the reference does)"""
    import random

    import sentencepiece as spm
    rng = random.Random(5)
    syl = [c + v for c in "bdfghklmnprstvz" for v in "aeiou"]
    words = ["hello", "my", "name", "is"] + ["".join(rng.choice(syl) for _ in range(rng.randint(1, 4))) for _ in range(600)]
    corpus = dirpath / "corpus.txt"
    with open(corpus, "w") as f:
        for _ in range(1500):
            f.write(" ".join(rng.choice(words) for _ in range(12)) + "\n")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(dirpath / "tokenizer"), vocab_size=vocab, model_type="bpe",
                                   bos_id=1, eos_id=2, unk_id=0, pad_id=-1, minloglevel=2)
    sp = spm.SentencePieceProcessor(str(dirpath / "tokenizer.model"))
    assert sp.vocab_size() == vocab and len(sp.DecodeIds(list(range(vocab)))) > 0


def _args(G, **kw):
    a = G.build_parser().parse_args([])
    for k, v in kw.items():
        assert hasattr(a, k), k
        setattr(a, k, v)
    return a


def test_checkpoint_and_histogram_flow(tmp_path, capsys):
    from teal_amd import calibrate
    from teal_amd.distribution import Distribution
    from teal_amd.gpt_fast import generate as G
    from teal_amd.monkeypatch import PROJ_HIST
    from teal_amd.quantize import quantize_model_int8
    ck = tmp_path / "checkpoints" / "tiny-test"
    ck.mkdir(parents=True)
    model = G.build_synthetic_model("tiny-test", DEV, torch.float16, seed=21, std=0.05)
    torch.save(model.state_dict(), ck / "model.pth")
    _tokenizer_model(ck)
    # calibration: histograms of the block inputs / intermediates on a batch of token ids (teal/grab_acts.py)
    teal_path = tmp_path / "teal"
    ids = torch.randint(0, model.config.vocab_size, (4, 64), generator=torch.Generator().manual_seed(8))
    calibrate.grab_histograms(model, ids, str(teal_path), save_activations=False)
    hist = teal_path / "histograms"
    n_layer = len(model.layers)
    assert all((hist / f"layer-{i}" / sub / "histograms.pt").is_file() for i in range(n_layer) for sub in ("self_attn", "mlp"))
    int8_sd = quantize_model_int8(G.build_synthetic_model("tiny-test", DEV, torch.float16, seed=21, std=0.05)).state_dict()
    torch.save(int8_sd, ck / "model_int8.pth")
    del model
    s = 0.5
    common = dict(checkpoint_path=ck / "model.pth", hist_path=str(hist), sparsity=s, num_samples=1, max_new_tokens=10, top_k=1,
                  temperature=1.0, prompt="hello my name is", device=DEV)

    # (a) the flow as the reference user types it: --compile => the device-resident engine loop
    ra = G.main(_args(G, compile=True, **common))
    assert ra["decoder"] == "EngineDecoder"
    # (i) thresholds = icdf(0.5 + 0.5 s) of the layer's histogram (gpt-fast/generate.py:277-287)
    for i, th in enumerate(ra["thresholds"]):
        for proj, (sub, h) in PROJ_HIST.items():
            want = Distribution(str(hist / f"layer-{i}" / sub), h).icdf(0.5 + 0.5 * s).item()
            assert th[proj] == want, (i, proj)
    assert len(ra["thresholds"]) == n_layer and ra["thresholds"][0]["q"] > 0

    # (b) the reference harness's own loop over the patched model (model(token, pos) + torch sampler in a hipGraph)
    rb = G.main(_args(G, compile=True, no_engine=True, **common))
    # (c) op-by-op module path (torch.ops.teal.* + eager glue), eager
    rc = G.main(_args(G, compile=False, no_fused_decode=True, **common))
    assert rb["decoder"] == "GraphedDecoder" and rc["decoder"] == "GraphedDecoder"
    assert ra["thresholds"] == rb["thresholds"] == rc["thresholds"]
    # (ii) greedy (top_k = 1) tokens.  The engine loop and the harness loop run the same fused launches: identical.  The
    # op-by-op path computes RMSNorm / silu with torch ops, so an activation within an ulp of a threshold can fall on the
    # other side (tests/test_engine.py bounds the logits: cosine > 0.995 at 50 %): the greedy continuation agrees until
    # such a flip tips an argmax — the first three generated tokens at s = 0.5, all of them at s = 0
    sa, sb, sc = ra["sequences"][0], rb["sequences"][0], rc["sequences"][0]
    n_prompt = len(sa) - 10
    assert sa == sb, (sa, sb)
    assert len(sa) == len(sc) > 10 and sa[: n_prompt + 3] == sc[: n_prompt + 3], (sa, sc)
    zero = dict(common, sparsity=0.0)
    za = G.main(_args(G, compile=True, **zero))
    zc = G.main(_args(G, compile=False, no_fused_decode=True, **zero))
    assert za["sequences"][0] == zc["sequences"][0], (za["sequences"][0], zc["sequences"][0])

    # (iii) an *int8* checkpoint name takes the int8 weight-only branch (gpt-fast/generate.py:236-243)
    capsys.readouterr()
    ri = G.main(_args(G, compile=True, **dict(common, checkpoint_path=ck / "model_int8.pth")))
    out = capsys.readouterr().out
    assert "Using int8 weight-only quantization!" in out and "Monkeypatching with activation sparsity" in out
    assert ri["decoder"] == "EngineDecoder" and ri["thresholds"] == ra["thresholds"]
    assert len(ri["sequences"][0]) == len(ra["sequences"][0])
    json.dumps(ri["sequences"])  # plain ints

    # (iv) an *int4*.g<G>.pth checkpoint (written by teal_amd.quantize.quantize_model_int4; the reference's own int4 files
    # hold a CUDA-only packed layout) takes the int4 branch with the group size parsed from the name (generate.py:236-242)
    from teal_amd.quantize import quantize_model_int4
    q4 = quantize_model_int4(G.build_synthetic_model("tiny-test", DEV, torch.float16, seed=21, std=0.05), 32)
    torch.save(q4.state_dict(), ck / "model_int4.g32.pth")
    del q4
    capsys.readouterr()
    r4 = G.main(_args(G, compile=True, **dict(common, checkpoint_path=ck / "model_int4.g32.pth")))
    out = capsys.readouterr().out
    assert "Using int4 weight-only quantization!" in out and r4["thresholds"] == ra["thresholds"]
    assert r4["decoder"] == "EngineDecoder"  # the fused engine over int4 blocks
    assert len(r4["sequences"][0]) == len(ra["sequences"][0])


def test_monkeypatch_layer_llama2_7b_width_with_committed_histograms():
    """monkeypatch_layer on a Llama-2-7B-width block with the reference's own Llama-2-7B histograms (layers 0 and 15,
    fixture F6): thresholds equal fixture F1 (the reference's Distribution.icdf), and the patched single-token forwards —
    the HIP sparse GEMVs — agree with a masked-matmul restatement of gpt-fast/model.py:163-190,258-259 at those
    thresholds."""
    from teal_amd.gpt_fast.model import ModelArgs, TransformerBlock, precompute_freqs_cis, KVCache
    from teal_amd.monkeypatch import monkeypatch_layer
    with open(os.path.join(GOLDEN, "thresholds.json")) as f:
        F1 = json.load(f)
    i5 = F1["levels"].index(0.5)
    cfg = ModelArgs.from_name("Llama-2-7b-chat-hf")
    g = torch.Generator(device=DEV).manual_seed(31)
    for li in (0, 15):
        with torch.device(DEV):
            layer = TransformerBlock(cfg).half()
        with torch.no_grad():
            for p in layer.parameters():
                p.copy_(torch.randn(p.shape, device=DEV, generator=g) * 0.02)
        dense = {n: getattr(m, n).weight.detach().clone() for m, names in ((layer.attention, ("wqkv", "wo")), (layer.feed_forward, ("w1", "w3", "w2")))
                 for n in names}
        th = monkeypatch_layer(li, layer, 0.5, HIST, DEV)
        want = F1["models"]["Llama-2-7B"][li]
        assert (th["q"], th["k"], th["v"], th["o"]) == (want["attn_h1"][i5],) * 3 + (want["attn_h2"][i5],)
        assert (th["gate"], th["up"], th["down"]) == (want["mlp_h1"][i5],) * 2 + (want["mlp_h2"][i5],)
        at, ff = layer.attention, layer.feed_forward
        assert (at.thresh_q, at.thresh_o, ff.thresh_gate, ff.thresh_down) == (th["q"], th["o"], th["gate"], th["down"])
        # activations on the scale of the real model's (the histograms describe them): a few rows pass, most do not
        x = (torch.randn(1, 1, cfg.dim, device=DEV, generator=g) * 1.5 * max(th["gate"], 1e-3)).half()
        mask = lambda v, t: torch.where(v.float().abs() > t, v, torch.zeros_like(v))  # noqa: E731
        tol = lambda ref: 1e-3 * max(1.0, float(ref.abs().max())) + float(ref.abs().max()) * 2.0 ** -10  # noqa: E731  (+ 1 ulp fp16)
        with torch.no_grad():
            # stage by stage, each against a masked matmul of the SAME input bits (so no activation can sit on the other
            # side of a threshold in the restatement)
            kept = float((x.float().abs() > th["gate"]).float().mean())
            assert 0.05 < kept < 0.95
            gate = ff.gemv1(x, ff.w1.weight, ff.thresh_gate, ff.sparsity_bin)
            up = ff.gemv1(x, ff.w3.weight, ff.thresh_up, ff.sparsity_bin)
            for got_p, wname, t in ((gate, "w1", th["gate"]), (up, "w3", th["up"])):
                ref = mask(x, t).float() @ dense[wname].float().T
                assert (got_p.float() - ref).abs().max() <= tol(ref), (li, wname)
            hmid = torch.nn.functional.silu(gate) * up
            got = ff(x)  # the patched forward (model.py:258-259): the same three ops
            ref = mask(hmid, th["down"]).float() @ dense["w2"].float().T
            assert (got.float() - ref).abs().max() <= tol(ref), (li, "w2")
            assert torch.equal(got, ff.gemv2(hmid, ff.w2.weight, ff.thresh_down, ff.sparsity_bin))
            # attention side: q|k|v with tau from attn_h1, o with attn_h2 (single token at position 0: softmax over one key,
            # so the attention output is v; n_local_heads == n_head for Llama-2-7B)
            at.kv_cache = KVCache(1, 8, cfg.n_local_heads, cfg.head_dim, torch.float16).to(DEV)
            fc = precompute_freqs_cis(8, cfg.head_dim, cfg.rope_base, torch.float16).to(DEV)
            pos = torch.tensor([0], device=DEV)
            xa = (torch.randn(1, 1, cfg.dim, device=DEV, generator=g) * 1.5 * max(th["q"], 1e-3)).half()
            cm = torch.zeros(1, 1, 1, 8, dtype=torch.bool, device=DEV)
            cm[..., 0] = True
            qkv = at.gemv1(xa, at.wqkv.weight, at.thresh_q, at.thresh_k, at.thresh_v, at.sparsity_bin, cfg.n_local_heads * cfg.head_dim)
            ref = mask(xa, th["q"]).float() @ dense["wqkv"].float().T
            assert (qkv.float() - ref).abs().max() <= tol(ref), (li, "wqkv")
            got_a = at(xa, fc[pos], cm, pos)
            v = qkv[..., 2 * cfg.dim:]
            ref_a = mask(v, th["o"]).float() @ dense["wo"].float().T
            assert (got_a.float() - ref_a).abs().max() <= tol(ref_a), (li, "wo")
        del layer, dense
        torch.cuda.empty_cache()
