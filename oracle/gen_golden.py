#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own code in this container.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (the build container);
its outputs (small data fixtures) are committed, the reference's Python never travels.

What is executed from the reference (read-only import, nothing copied):
  * kernels/sparse_gemv.py  splitk_sparse_gemv_kernel / qkv_kernel  — under TRITON_INTERPRET=1,
    through `.fn[grid](...)` (the Autotuner needs a GPU driver to benchmark; its pre_hook's
    Y.zero_() is done here by the caller).  `msgspec` is absent from the image and only used
    for a dataclass base, so a 5-line stand-in module is injected before the import.
  * gpt-fast/distribution.py  Distribution.icdf      — thresholds (F1)
  * utils/utils.py  get_layer_greedy_sparsities       — block-wise greedy tables (F2)
  * utils/utils.py  SparsifyFn.apply                  — the fp16-compare mask rule (F5b)

Fixtures written:
  F1 thresholds.json         tau for every layer x {attn_h1,attn_h2,mlp_h1,mlp_h2} x sparsity
  F2 greedy_llama2_7b.json   per-layer/per-projection greedy sparsities + thresholds
  F3 kat_gemv_*.npz          x, tau, kept idx, y_ref (two tilings), y_truth64; W from hash_uniform
  F4 kat_qkv_*.npz           same with three thresholds (MHA + GQA geometries)
  F5 kat_boundary.npz        compare-rule edge values run through the reference kernel
  F6 hist/…/histograms.pt, lookup/…/results.csv   raw calibration data files (MIT, data only)
  F7 kat_hist_producer.npz   ActivationModule.find_histogram on seeded activations (producer-side KAT)
  F10 kat_int4.npz          group_quantize_tensor / group_dequantize_tensor (gpt-fast/quantize.py:58-162) for G = 32, 128
  F9 greedy_driver/          teal/greedyopt.py process_layer (the greedy driver + the reference's SparsifyFn wiring) run
                             on a tiny seeded block: model.pt, histograms/, activations/, lookup/layer-i/results.csv

Usage:  python oracle/gen_golden.py [--quick]
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import time
import types

os.environ["TRITON_INTERPRET"] = "1"

import numpy as np
import torch
import triton

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import teal_oracle as O  # noqa: E402


def import_reference():
    ms = types.ModuleType("msgspec")

    class Struct:  # stand-in for msgspec.Struct (positional dataclass)
        def __init__(self, *a):
            for k, v in zip(self.__annotations__, a):
                setattr(self, k, v)

    ms.Struct = Struct
    sys.modules["msgspec"] = ms
    sys.path.insert(0, REF)
    from kernels.sparse_gemv import qkv_kernel, splitk_sparse_gemv_kernel  # type: ignore
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_distribution", os.path.join(REF, "gpt-fast", "distribution.py"))
    dist = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dist)
    from utils.utils import SparsifyFn, get_layer_greedy_sparsities  # type: ignore

    return splitk_sparse_gemv_kernel.fn, qkv_kernel.fn, dist.Distribution, get_layer_greedy_sparsities, SparsifyFn


def t16(bits: np.ndarray, dtype: int) -> torch.Tensor:
    t = torch.from_numpy(bits.copy().view(np.int16))
    return t.view(torch.float16 if dtype == O.F16 else torch.bfloat16)


def run_ref_gemv(inner, xb, wb, tau, Z, N, dtype, BM, BN):
    x = t16(xb, dtype).view(1, 1, Z)
    W = t16(wb, dtype).view(Z, N).T  # [N, Z], strides (1, N): the reference's "column major"
    assert W.stride(1) > 1
    y = torch.zeros(1, 1, N, dtype=torch.float16)  # pre_hook init_to_zero("Y")
    grid = (triton.cdiv(N, BN), triton.cdiv(Z, BM))
    inner[grid](y, W, x, tau, N, Z, N // 16, Z // 16, BATCHSIZE=1, SPARSITY_BIN=0, BLOCK_N=BN, BLOCK_M=BM)
    return y.view(-1).numpy().view(np.uint16).copy()


def run_ref_qkv(inner, xb, wb, tq, tk, tv, Z, N, N_q, N_kv, dtype, BM, BN):
    x = t16(xb, dtype).view(1, 1, Z)
    W = t16(wb, dtype).view(Z, N).T
    y = torch.zeros(1, 1, N, dtype=torch.float16)
    grid = (triton.cdiv(N, BN), triton.cdiv(Z, BM))
    inner[grid](y, W, x, tq, tk, tv, N, N_q, N_kv, Z, N // 16, Z // 16, BATCHSIZE=1, SPARSITY_BIN=0,
                BLOCK_N=BN, BLOCK_M=BM)
    return y.view(-1).numpy().view(np.uint16).copy()


TILINGS = [(128, 512), (16, 256)]


def gen_thresholds(Distribution, quick):
    models = {"Llama-2-7B": 32, "Llama-3-8B": 32, "Llama-2-70B": 80}
    levels = [0.0, 0.1, 0.2, 0.25, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
    out = {"levels": levels, "rule": "tau = Distribution(path, h).icdf(0.5 + 0.5*s).item()  (gpt-fast/generate.py:277-287)",
           "models": {}}
    for name, L in models.items():
        layers = []
        for i in range(L if not quick else 2):
            row = {}
            for tag, sub, h in (("attn_h1", "self_attn", "h1"), ("attn_h2", "self_attn", "h2"),
                                ("mlp_h1", "mlp", "h1"), ("mlp_h2", "mlp", "h2")):
                d = Distribution(os.path.join(REF, "models", name, "histograms", f"layer-{i}", sub), h)
                row[tag] = [d.icdf(0.5 + 0.5 * s).item() for s in levels]
            layers.append(row)
        out["models"][name] = layers
        print(f"  F1 {name}: {len(layers)} layers")
    with open(os.path.join(OUT, "thresholds.json"), "w") as f:
        json.dump(out, f)


def gen_greedy(Distribution, get_greedy):
    name, L = "Llama-2-7B", 32
    lookup = os.path.join(REF, "models", name, "lookup")
    out = {"model": name, "targets": {}}
    proj_hist = {"q": ("self_attn", "h1"), "k": ("self_attn", "h1"), "v": ("self_attn", "h1"),
                 "o": ("self_attn", "h2"), "gate": ("mlp", "h1"), "up": ("mlp", "h1"), "down": ("mlp", "h2")}
    dists = {}
    for t in (0.3, 0.4, 0.5, 0.6):
        sp = get_greedy([t] * L, lookup)
        sp = {p: [float(v) for v in vals] for p, vals in sp.items()}
        taus = {p: [] for p in sp}
        for p, vals in sp.items():
            sub, h = proj_hist[p]
            for i, s in enumerate(vals):
                key = (i, sub, h)
                if key not in dists:
                    dists[key] = Distribution(os.path.join(REF, "models", name, "histograms", f"layer-{i}", sub), h)
                taus[p].append(dists[key].icdf(0.5 + 0.5 * s).item())
        out["targets"][repr(t)] = {"sparsities": sp, "thresholds": taus}
        print(f"  F2 target {t}: layer0 q={sp['q'][0]:.3f} down={sp['down'][0]:.3f}")
    with open(os.path.join(OUT, "greedy_llama2_7b.json"), "w") as f:
        json.dump(out, f)


def gen_gemv_kats(inner, quick):
    # (tag, Z, N, dtype, scale_x, scale_w, tau, store_W)
    cases = [
        ("small_f16", 256, 512, O.F16, 1.0, 1.0, 0.25, True),
        ("small_bf16", 256, 512, O.BF16, 1.0, 1.0, 0.20, True),
        ("wo_7b_f16", 4096, 4096, O.F16, 1.0, 1.0, 0.25, False),          # benchmark law: tau = s/2, s = .5
        ("gate_7b_f16", 4096, 11008, O.F16, 4.0, 0.08, 1.0, False),        # ~unit-scale outputs
        ("down_7b_f16", 11008, 4096, O.F16, 2.0, 0.08, 0.5, False),
        ("gate_8b_bf16", 4096, 14336, O.BF16, 4.0, 0.08, 0.8, False),      # 40 % sparsity, bf16 inputs
    ]
    if quick:
        cases = cases[:2]
    for tag, Z, N, dtype, sx, sw, tau, store_w in cases:
        t0 = time.time()
        xb = O.hash_uniform(Z, seed=1000 + Z + N, scale=sx, dtype=dtype)
        wb = O.hash_uniform_c(Z * N, seed=2000 + Z + N, scale=sw, dtype=dtype)
        idx = O.compact(xb, tau, dtype)
        refs = {}
        for BM, BN in TILINGS:
            if N % BN:
                continue
            refs[f"y_ref_{BM}x{BN}"] = run_ref_gemv(inner, xb, wb, tau, Z, N, dtype, BM, BN)
        truth = O.truth64(xb, wb, Z, N, tau, dtype=dtype)
        payload = dict(x=xb, tau=np.float32(tau), Z=Z, N=N, dtype=dtype, seed_w=2000 + Z + N, scale_w=np.float32(sw),
                       kept=idx, y_truth64=truth, **refs)
        if store_w:
            payload["wT"] = wb
        np.savez_compressed(os.path.join(OUT, f"kat_gemv_{tag}.npz"), **payload)
        err = {k: float(np.abs(O.from_bits(v, O.F16).astype(np.float64) - truth).max()) for k, v in refs.items()}
        print(f"  F3 {tag}: nnz={idx.size}/{Z} ref max|err| {err} ({time.time() - t0:.1f}s)")


def gen_index_only(quick):
    # 8192-wide (70B) index-only KATs: no GEMV, just the keep rule on bigger vectors
    for tag, Z, dtype, tau in (("z8192_f16", 8192, O.F16, 0.31), ("z28672_f16", 28672, O.F16, 0.17),
                               ("z28672_bf16", 28672, O.BF16, 0.25)):
        xb = O.hash_uniform(Z, seed=77 + Z, scale=1.0, dtype=dtype)
        v = O.from_bits(xb, dtype)
        # independent statement of the rule through torch fp32 (what Triton's compare does)
        kept = torch.nonzero(torch.from_numpy(v).abs() > torch.tensor(tau, dtype=torch.float32)).view(-1).numpy().astype(np.int32)
        assert np.array_equal(kept, O.compact(xb, tau, dtype))
        np.savez_compressed(os.path.join(OUT, f"kat_index_{tag}.npz"), x=xb, tau=np.float32(tau), dtype=dtype, kept=kept)
        print(f"  F3 index {tag}: nnz={kept.size}/{Z}")


def gen_qkv_kats(inner, quick):
    cases = [
        ("small_f16", 256, 768, 256, 256, O.F16, 1.0, 1.0, (0.10, 0.25, 0.40), True),
        ("mha_7b_f16", 4096, 12288, 4096, 4096, O.F16, 4.0, 0.08, (1.6, 1.2, 0.6), False),   # 7B: kv_size 4096
        ("gqa_8b_bf16", 4096, 6144, 4096, 1024, O.BF16, 4.0, 0.08, (0.5, 1.0, 1.5), False),  # 8B: kv_size 1024
    ]
    if quick:
        cases = cases[:1]
    for tag, Z, N, N_q, N_kv, dtype, sx, sw, (tq, tk, tv), store_w in cases:
        t0 = time.time()
        xb = O.hash_uniform(Z, seed=3000 + Z + N, scale=sx, dtype=dtype)
        wb = O.hash_uniform_c(Z * N, seed=4000 + Z + N, scale=sw, dtype=dtype)
        refs = {}
        for BM, BN in TILINGS:
            if N % BN or N_q % BN or N_kv % BN:
                continue
            refs[f"y_ref_{BM}x{BN}"] = run_ref_qkv(inner, xb, wb, tq, tk, tv, Z, N, N_q, N_kv, dtype, BM, BN)
        if not refs:  # small case: use a tiling that divides the segments
            refs["y_ref_64x128"] = run_ref_qkv(inner, xb, wb, tq, tk, tv, Z, N, N_q, N_kv, dtype, 64, 128)
        truth = O.truth64(xb, wb, Z, N, tq, tk, tv, N_q, N_kv, dtype)
        payload = dict(x=xb, tau_q=np.float32(tq), tau_k=np.float32(tk), tau_v=np.float32(tv), Z=Z, N=N, N_q=N_q,
                       N_kv=N_kv, dtype=dtype, seed_w=4000 + Z + N, scale_w=np.float32(sw),
                       kept_q=O.compact(xb, tq, dtype), kept_k=O.compact(xb, tk, dtype),
                       kept_v=O.compact(xb, tv, dtype), y_truth64=truth, **refs)
        if store_w:
            payload["wT"] = wb
        np.savez_compressed(os.path.join(OUT, f"kat_qkv_{tag}.npz"), **payload)
        err = {k: float(np.abs(O.from_bits(v, O.F16).astype(np.float64) - truth).max()) for k, v in refs.items()}
        print(f"  F4 {tag}: ref max|err| {err} ({time.time() - t0:.1f}s)")


def gen_boundary(inner, SparsifyFn):
    """Edge values of the compare rule, decided by the reference kernel itself.

    W = identity (Z = N = 64), so y[m] = x[m] if row m is kept, else 0 — except that a NaN in x
    poisons every column of the reference's output (masked rows still multiply 0 * NaN).
    """
    Z = N = 64
    f16 = lambda v: np.array(v, dtype=np.float16)  # noqa: E731
    tau = 0.09997  # the survey's probe: fp16(0.1) = 0.09997559 > 0.09997 in fp32, but fp16(tau) == fp16(0.1)
    vals = np.zeros(Z, dtype=np.float16)
    special = [0.1, -0.1, 0.0999, 0.09991, 0.0, -0.0, 6.0e-8, -6.0e-8, 65504.0, -65504.0, 3.0e-5, -3.0e-5,
               0.09985, 0.1001, 1.0, -1.0, 5.96e-8, 0.099976, -0.099976, 0.09992]
    vals[: len(special)] = f16(special)
    vals[len(special):] = (np.linspace(-0.2, 0.2, Z - len(special))).astype(np.float16)
    xb = vals.view(np.uint16).copy()
    eye = np.eye(Z, dtype=np.float16).view(np.uint16).reshape(-1).copy()
    out = {}
    for name, t in (("tau_probe", tau), ("tau_zero", 0.0), ("tau_tiny", 1e-10), ("tau_exact_x", float(np.float16(0.1))),
                    ("tau_below_x", float(np.float32(np.float16(0.1)) - np.float32(1e-10)))):
        y = run_ref_gemv(inner, xb, eye, t, Z, N, O.F16, 16, 16)
        kept = np.nonzero(O.from_bits(y, O.F16) != 0)[0].astype(np.int32)  # zero inputs can never show as kept
        out[f"{name}_tau"] = np.float32(t)
        out[f"{name}_y"] = y
        out[f"{name}_kept_nonzero"] = kept
    # NaN: poisons the whole reference output
    xn = xb.copy()
    xn[5] = np.array([np.nan], dtype=np.float16).view(np.uint16)[0]
    out["nan_x"] = xn
    out["nan_y"] = run_ref_gemv(inner, xn, eye, tau, Z, N, O.F16, 16, 16)
    # +inf is always kept; inf * 0 (identity's zeros) makes every other column NaN
    xi = xb.copy()
    xi[7] = np.array([np.inf], dtype=np.float16).view(np.uint16)[0]
    out["inf_x"] = xi
    out["inf_y"] = run_ref_gemv(inner, xi, eye, tau, Z, N, O.F16, 16, 16)
    # the fp16-compare rule of SparsifyFn.apply on the same vector (utils/utils.py:51-52)
    class _D:  # minimal stand-in distribution; set_threshold path not used
        def icdf(self, q):
            return torch.tensor(0.0)
    fn = SparsifyFn(_D())
    fn.threshold = tau
    masked = fn.apply(torch.from_numpy(vals.copy()).view(1, 1, Z)).view(-1).numpy().view(np.uint16).copy()
    out["sparsifyfn_tau"] = np.float32(tau)
    out["sparsifyfn_out"] = masked
    np.savez_compressed(os.path.join(OUT, "kat_boundary.npz"), x=xb, **out)
    k = out["tau_probe_kept_nonzero"]
    print(f"  F5 boundary: probe keeps fp16(0.1)? {0 in k};  nan poisons all: {bool(np.isnan(O.from_bits(out['nan_y'], 0)).all())}")


def gen_hist_producer():
    """F7: the reference's own ActivationModule.find_histogram (utils/utils.py:145-177) on a seeded
    activation tensor.  It moves data to 'cuda' for the sort; there is no GPU in this container, so
    Tensor.to is shimmed to ignore that one device argument while the reference code runs."""
    from utils.utils import ActivationModule  # type: ignore
    g = torch.Generator().manual_seed(123)
    acts = torch.randn(6, 512, generator=g) * 0.7
    acts[0, :8] = torch.tensor([9.0, -11.0, 7.5, -6.0, 12.0, -3.0, 5.0, 4.0])  # outliers
    am = ActivationModule("/tmp/unused")
    am.activations = {"h1": [acts[:3]], "h2": [acts[3:] * 2.0 + 0.1]}
    orig_to = torch.Tensor.to

    def to_no_cuda(self, *a, **k):
        if a and a[0] == "cuda":
            return self
        return orig_to(self, *a, **k)

    torch.Tensor.to = to_no_cuda
    try:
        hist = am.find_histogram(num_bins=1000, outlier_threshold=0.01)
    finally:
        torch.Tensor.to = orig_to
    np.savez_compressed(os.path.join(OUT, "kat_hist_producer.npz"), acts=acts.numpy(), num_bins=1000,
                        **{k: v.numpy() for k, v in hist.items()})
    print("  F7 find_histogram:", {k: tuple(v.shape) for k, v in hist.items()})


def gen_greedy_driver():
    """F9: the reference's block-wise greedy optimiser DRIVER (teal/greedyopt.py:99-159 process_layer, with its own
    f(), step sizes, calculate_activation_error / calculate_baseline_error and CSV writer) and its own SparsifyFn /
    Distribution wiring (teal/mlp.py:14-35, teal/self_attn.py:21-44 _monkeypatch_* on holder modules), executed here
    on a tiny seeded gpt-fast-shaped block.  The reference's HF decoder-layer forward cannot run in this image
    (transformers 5.x changed LlamaDecoderLayer.forward's signature: 'takes from 2 to 7 positional arguments but 8 were
    given'), so the block arithmetic behind the driver is teal_amd.calibrate.layer_forward — the same on both sides of
    the test; what the fixture pins is every decision the optimiser takes and every number it writes."""
    import teal.greedyopt as RG  # type: ignore
    import teal.mlp as RM  # type: ignore
    import teal.self_attn as RS  # type: ignore
    from teal_amd.calibrate import _prefill_tables, grab_histograms, layer_forward
    from teal_amd.gpt_fast.model import ModelArgs, Transformer
    out = os.path.join(OUT, "greedy_driver")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    cfg = ModelArgs(block_size=64, vocab_size=128, n_layer=2, n_head=4, dim=64, intermediate_size=128, n_local_heads=2)
    torch.manual_seed(20240907)
    model = Transformer(cfg).eval()
    with torch.no_grad():
        for n, prm in model.named_parameters():
            prm.copy_(torch.ones_like(prm) if n.endswith("norm.weight") else torch.randn_like(prm) * 0.08)
    torch.save(model.state_dict(), os.path.join(out, "model.pt"))
    ids = torch.randint(0, cfg.vocab_size, (2, 48), generator=torch.Generator().manual_seed(5))
    grab_histograms(model, ids, out, num_bins=2000)
    meta = dict(config=dict(block_size=64, vocab_size=128, n_layer=2, n_head=4, dim=64, intermediate_size=128, n_local_heads=2),
                model_type="Llama-2-7B", target_sparsity=0.6, base_step_size=0.05, last_fraction=0.25, num_bins=2000)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.anchor = torch.nn.Parameter(torch.zeros(1))

    class Block(torch.nn.Module):
        """what process_layer needs from a decoder layer: .mlp.sparse_fns, .self_attn.sparse_fns, .to(), __call__"""

        def __init__(self, blk, i, fc, mask):
            super().__init__()
            self.blk, self.fc, self.mask = blk, fc, mask
            self.mlp = RM._monkeypatch_mlp(Holder(), os.path.join(out, "histograms", f"layer-{i}", "mlp"))
            self.self_attn = RS._monkeypatch_self_attn(Holder(), os.path.join(out, "histograms", f"layer-{i}", "self_attn"))

        def to(self, *a, **k):
            return self

        def forward(self, hidden_states, attention_mask, position_ids, past_key_value, output_attentions, use_cache, cache_position):
            sp = {**dict(self.self_attn.sparse_fns.items()), **dict(self.mlp.sparse_fns.items())}
            return (layer_forward(self.blk, hidden_states, self.fc, self.mask, sp),)

    orig_to = torch.Tensor.to

    def to_no_cuda(self, *a, **k):
        if a and a[0] == "cuda":
            return self
        return orig_to(self, *a, **k)

    torch.Tensor.to = to_no_cuda
    try:
        with torch.no_grad():
            for i, blk in enumerate(model.layers):
                acts = torch.load(os.path.join(out, "activations", f"act_{i}.pt"))
                fc, mask = _prefill_tables(model, acts.shape[1], acts.device)
                final = RG.process_layer(Block(blk, i, fc, mask), meta["model_type"], i, meta["target_sparsity"],
                                         meta["base_step_size"], meta["last_fraction"], out)
                meta[f"final_{i}"] = final
    finally:
        torch.Tensor.to = orig_to
    with open(os.path.join(out, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    rows = sum(1 for _ in open(os.path.join(out, "lookup", "layer-0", "results.csv"))) - 1
    print(f"  F9 greedy driver: {rows} steps for layer 0; final {meta['final_0']}")


def gen_int8():
    """F8: the reference's int8 weight-only quantiser and module (gpt-fast/quantize.py:24-56, :339-357), run here:
    q / scales of a seeded weight (incl. an all-zero row, a row whose extreme is negative, a row of tiny values),
    and the module's forward on the TEAL-masked activation (x where fp32(|x|) > fp32(tau), else 0) — the
    composition the int8 sparse GEMV implements.  `tiktoken` (absent here) is only imported by the reference's
    tokenizer module; an empty stand-in lets quantize.py import."""
    sys.modules.setdefault("tiktoken", types.ModuleType("tiktoken"))
    tl = types.ModuleType("tiktoken.load")
    tl.load_tiktoken_bpe = lambda *a, **k: {}
    sys.modules.setdefault("tiktoken.load", tl)
    gf = os.path.join(REF, "gpt-fast")
    if gf not in sys.path:
        sys.path.insert(0, gf)
    import quantize as RQ  # type: ignore

    out = {}
    for tag, dtype, N, Z in (("f16", O.F16, 96, 256), ("bf16", O.BF16, 64, 192)):
        tdt = torch.float16 if dtype == O.F16 else torch.bfloat16
        wb = O.hash_uniform_c(N * Z, 401 + dtype, 0.08, dtype)  # [Z][N] image; the module wants [N, Z]
        w = t16(wb, dtype).view(Z, N).T.contiguous().clone()
        w[3] = 0                      # all-zero row -> scale clamps to eps
        w[5] = -w[5].abs()            # extreme is negative
        w[7] = w[7] * 1e-3            # tiny values
        q, scales, _ = RQ.dynamically_quantize_per_channel(w.float(), -128, 127, torch.int8)
        mod = RQ.WeightOnlyInt8Linear(Z, N)
        mod.weight.copy_(q)
        mod.scales = scales.to(tdt)   # WeightOnlyInt8QuantHandler stores scales in the model dtype (quantize.py:330)
        xb = O.hash_uniform(Z, 402 + dtype, 2.0, dtype)
        x = t16(xb, dtype).view(1, 1, Z)
        tau = 0.7
        keep = x.float().abs() > torch.tensor(tau, dtype=torch.float32)
        xm = torch.where(keep, x, torch.zeros_like(x))
        y = mod(xm)
        yd = mod(x)
        assert y.dtype == tdt
        out[f"{tag}_w"] = w.view(torch.int16).numpy().view(np.uint16)
        out[f"{tag}_q"] = q.numpy()
        out[f"{tag}_scales_f32"] = scales.numpy()
        out[f"{tag}_scales"] = scales.to(tdt).view(torch.int16).numpy().view(np.uint16)
        out[f"{tag}_x"] = xb
        out[f"{tag}_tau"] = np.float32(tau)
        out[f"{tag}_y_masked"] = y.view(-1).view(torch.int16).numpy().view(np.uint16)
        out[f"{tag}_y_dense"] = yd.view(-1).view(torch.int16).numpy().view(np.uint16)
        print(f"  F8 int8 {tag}: N={N} Z={Z} kept={int(keep.sum())}")
    np.savez_compressed(os.path.join(OUT, "kat_int8.npz"), **out)


def gen_int4():
    """F10: the reference's int4 group quantiser (gpt-fast/quantize.py:58-162: get_group_qparams,
    group_quantize_tensor(+_from_qparams, pack_scales_and_zeros), group_dequantize_tensor), run here on seeded weights
    for group sizes 32 and 128: q, scales_and_zeros (bf16 [Z / G][N][2]), the dequantised weight, and — the composition
    the int4 sparse GEMV implements — y = masked(x) @ dequant(W).T in float64 for a 3-threshold and a 1-threshold case.
    (The reference module's own forward is a CUDA-only packed matmul, quantize.py:366-372; it cannot run here.)"""
    sys.modules.setdefault("tiktoken", types.ModuleType("tiktoken"))
    tl = types.ModuleType("tiktoken.load")
    tl.load_tiktoken_bpe = lambda *a, **k: {}
    sys.modules.setdefault("tiktoken.load", tl)
    gf = os.path.join(REF, "gpt-fast")
    if gf not in sys.path:
        sys.path.insert(0, gf)
    import quantize as RQ  # type: ignore

    out = {}
    for tag, G, N, Z, N_q, N_kv in (("g32", 32, 512, 1024, 256, 128), ("g128", 128, 128, 2048, 128, 0)):
        wb = O.hash_uniform_c(N * Z, 501 + G, 0.1, O.BF16)
        w = t16(wb, O.BF16).view(Z, N).T.contiguous().clone()  # [N, Z] bf16
        w[3, :G] = 0                       # an all-zero group -> scale clamps to 1e-6
        w[5, G:2 * G] = w[5, G:2 * G].abs() + 0.01   # an all-positive group
        q, sz = RQ.group_quantize_tensor(w, n_bit=4, groupsize=G)
        wdq = RQ.group_dequantize_tensor(q, sz.float(), 4, G)  # unpack_scales_and_zeros asserts float (quantize.py:96-98)
        xb = O.hash_uniform(Z, 502 + G, 2.0, O.BF16)
        x = t16(xb, O.BF16).float().view(Z)
        taus = (0.5, 0.7, 0.3) if N_kv else (0.6, 0.6, 0.6)
        cols = [(0, N_q, taus[0]), (N_q, N_q + N_kv, taus[1]), (N_q + N_kv, N, taus[2])]
        y = np.zeros(N, np.float64)
        for c0, c1, tau in cols:
            if c1 > c0:
                keep = x.abs() > torch.tensor(tau, dtype=torch.float32)
                xm = torch.where(keep, x, torch.zeros_like(x)).double()
                y[c0:c1] = (wdq[c0:c1].double() @ xm).numpy()
        out[f"{tag}_w"] = w.view(torch.int16).numpy().view(np.uint16)
        out[f"{tag}_q"] = q.numpy().astype(np.uint8)
        out[f"{tag}_sz"] = sz.view(torch.int16).numpy().view(np.uint16)
        out[f"{tag}_wdq_rows16"] = wdq[:16].float().numpy()  # a slice of the dequantised weight; y pins all of it
        out[f"{tag}_x"] = xb
        out[f"{tag}_taus"] = np.array(taus, np.float32)
        out[f"{tag}_shape"] = np.array([N, Z, G, N_q, N_kv])
        out[f"{tag}_y"] = y
        print(f"  F10 int4 {tag}: N={N} Z={Z} q range {int(q.min())}..{int(q.max())}")
    np.savez_compressed(os.path.join(OUT, "kat_int4.npz"), **out)


def copy_raw_data():
    for sub in ("mlp", "self_attn"):
        for layer in (0, 15):
            src = os.path.join(REF, "models", "Llama-2-7B", "histograms", f"layer-{layer}", sub, "histograms.pt")
            dst = os.path.join(OUT, "hist", "Llama-2-7B", f"layer-{layer}", sub)
            os.makedirs(dst, exist_ok=True)
            shutil.copyfile(src, os.path.join(dst, "histograms.pt"))
    for layer in (0, 31):
        src = os.path.join(REF, "models", "Llama-2-7B", "lookup", f"layer-{layer}", "results.csv")
        dst = os.path.join(OUT, "lookup", "Llama-2-7B", f"layer-{layer}")
        os.makedirs(dst, exist_ok=True)
        shutil.copyfile(src, os.path.join(dst, "results.csv"))
    print("  F6 raw histograms.pt (layers 0,15) + results.csv (layers 0,31) copied")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    O.build()
    inner, qkv_inner, Distribution, get_greedy, SparsifyFn = import_reference()
    steps = {
        "thresholds": lambda: gen_thresholds(Distribution, a.quick),
        "greedy": lambda: gen_greedy(Distribution, get_greedy),
        "boundary": lambda: gen_boundary(inner, SparsifyFn),
        "index": lambda: gen_index_only(a.quick),
        "gemv": lambda: gen_gemv_kats(inner, a.quick),
        "qkv": lambda: gen_qkv_kats(qkv_inner, a.quick),
        "hist_producer": gen_hist_producer,
        "int8": gen_int8,
        "greedy_driver": gen_greedy_driver,
        "int4": gen_int4,
        "raw": copy_raw_data,
    }
    for name, fn in steps.items():
        if a.only and name not in a.only.split(","):
            continue
        print(f"[gen_golden] {name}")
        fn()
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "torch": torch.__version__, "triton": triton.__version__,
                   "numpy": np.__version__, "reference": "FasterDecoding/TEAL @ 2024-10-22 (/root/reference)",
                   "tilings": TILINGS}, f, indent=1)


if __name__ == "__main__":
    main()
