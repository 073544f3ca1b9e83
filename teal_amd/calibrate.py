"""Calibration producers: the offline tools that create the `histograms.pt`
and `lookup/layer-i/results.csv` files the decode path consumes.

  find_histogram            <- utils/utils.py:145-177 (ActivationModule.find_histogram)
  grab_histograms           <- teal/grab_acts.py:50-96 (layer-by-layer activation capture + histograms)
  greedy_optimize           <- teal/greedyopt.py:99-159 (block-wise greedy per-projection sparsities)

The reference drives HF-transformers models; here the same procedures run on the gpt-fast-shaped
Transformer of this repo (teal_amd/gpt_fast/model.py), so that a model with no calibration files —
e.g. the synthetic ones used on the GPU box — can be onboarded without the reference:

    python -m teal_amd.calibrate --synthetic tiny-test --output_path /tmp/teal_tiny --greedy
    python -m teal_amd.gpt_fast.generate --synthetic tiny-test --hist_path /tmp/teal_tiny/histograms --sparsity 0.5 ...

File formats are the reference's: histograms.pt = dict of fp32 [num_bins] tensors h1, h1_centers, h2,
h2_centers per `layer-i/{mlp,self_attn}`; results.csv header `Effective Sparsity,Activation Error,
Baseline Error,q,k,v,o,gate,up,down` (teal/greedyopt.py:123).  Activation sites: self_attn h1 = block
input after attention_norm (q/k/v), h2 = attention output before wo; mlp h1 = after ffn_norm (gate/up),
h2 = silu(gate)*up (down) — teal/self_attn.py, teal/mlp.py.
"""
from __future__ import annotations

import argparse
import csv
import os
from copy import deepcopy
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .distribution import Distribution
from .gpt_fast.model import Transformer, TransformerBlock, apply_rotary_emb
from .utils import PROJS, SparsifyFn

# relative parameter counts per projection (teal/greedyopt.py:26-52)
WEIGHT_DICT = {
    "Llama-3-8B": dict(q=1, k=1 / 4, v=1 / 4, o=1, gate=3.5, up=3.5, down=3.5),
    "Llama-3-70B": dict(q=1, k=1 / 8, v=1 / 8, o=1, gate=3.5, up=3.5, down=3.5),
    "Llama-2-7B": dict(q=1, k=1 / 8, v=1 / 8, o=1, gate=2.6875, up=2.6875, down=2.6875),
    "Llama-2-13B": dict(q=1, k=1 / 8, v=1 / 8, o=1, gate=2.7, up=2.7, down=2.7),
    "Llama-2-70B": dict(q=1, k=1 / 8, v=1 / 8, o=1, gate=3.5, up=3.5, down=3.5),
    "Mistral-7B": dict(q=1, k=1 / 8, v=1 / 8, o=1, gate=3.5, up=3.5, down=3.5),
}


def weights_from_config(cfg) -> Dict[str, float]:
    """projection sizes relative to q (what the table above encodes), derived from the architecture."""
    kv = cfg.n_local_heads * cfg.head_dim / cfg.dim
    mlp = cfg.intermediate_size / cfg.dim
    return dict(q=1.0, k=kv, v=kv, o=1.0, gate=mlp, up=mlp, down=mlp)


def find_histogram(acts: torch.Tensor, num_bins: int = 10000, outlier_threshold: float = 0.01):
    """(counts, bin_centers) of one activation site, the reference's binning exactly: sort; the main
    num_bins-2 bins span the [1 %, 99 %] quantile range uniformly; the first and last bin collect the
    outliers out to the min / max (utils/utils.py:155-172)."""
    a = torch.sort(acts.flatten().detach().float())[0]
    n = len(a)
    lower = a[int(outlier_threshold * n)]
    upper = a[-int(outlier_threshold * n)]
    a = a.cpu()
    main_bins = torch.linspace(lower, upper, num_bins - 1)
    bins = torch.cat([torch.tensor([a[0]]), main_bins, torch.tensor([a[-1]])])
    counts, _ = torch.histogram(a, bins=bins)
    centers = (bins[:-1] + bins[1:]) / 2
    return counts.float().cpu(), centers.float().cpu()


def _attn_parts(at, x, freqs_cis, mask, sp=None):
    """attention forward with optional per-projection SparsifyFns on the inputs of q/k/v and o; returns
    (output, h2) where h2 is the input of wo."""
    bsz, seqlen, _ = x.shape
    kv = at.n_local_heads * at.head_dim
    wq, wk, wv = at.wqkv.weight.split([at.dim, kv, kv], dim=0)
    xq, xk, xv = (sp["q"](x), sp["k"](x), sp["v"](x)) if sp else (x, x, x)
    q = F.linear(xq, wq).view(bsz, seqlen, at.n_head, at.head_dim)
    k = F.linear(xk, wk).view(bsz, seqlen, at.n_local_heads, at.head_dim)
    v = F.linear(xv, wv).view(bsz, seqlen, at.n_local_heads, at.head_dim)
    q, k = apply_rotary_emb(q, freqs_cis).transpose(1, 2), apply_rotary_emb(k, freqs_cis).transpose(1, 2)
    v = v.transpose(1, 2)
    rep = at.n_head // at.n_local_heads
    if rep > 1:
        k, v = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
    y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0)
    h2 = y.transpose(1, 2).contiguous().view(bsz, seqlen, at.dim)
    return at.wo(sp["o"](h2) if sp else h2), h2


def layer_forward(layer: TransformerBlock, x, freqs_cis, mask, sp=None, record=None):
    """one TransformerBlock on a [B, S, dim] prefill batch (no KV cache), optionally sparsified
    (teal/self_attn.py:46-156, teal/mlp.py:37-57) and/or recording the four activation sites."""
    h1a = layer.attention_norm(x)
    a_out, h2a = _attn_parts(layer.attention, h1a, freqs_cis, mask, sp)
    h = x + a_out
    h1m = layer.ffn_norm(h)
    ff = layer.feed_forward
    g = ff.w1(sp["gate"](h1m) if sp else h1m)
    u = ff.w3(sp["up"](h1m) if sp else h1m)
    h2m = F.silu(g) * u
    out = h + ff.w2(sp["down"](h2m) if sp else h2m)
    if record is not None:
        record.update(attn_h1=h1a, attn_h2=h2a, mlp_h1=h1m, mlp_h2=h2m)
    return out


def _prefill_tables(model: Transformer, seq_len: int, device):
    from .gpt_fast.model import precompute_freqs_cis
    c = model.config
    fc = precompute_freqs_cis(seq_len, c.head_dim, c.rope_base, model.output.weight.dtype).to(device)
    mask = torch.tril(torch.ones(seq_len, seq_len, dtype=torch.bool, device=device))[None, None]
    return fc, mask


@torch.no_grad()
def grab_histograms(model: Transformer, input_ids: torch.Tensor, output_path: str, num_bins: int = 10000,
                    save_activations: bool = True) -> None:
    """teal/grab_acts.py: run the calibration batch layer by layer, save each layer's INPUT hidden states
    (`activations/act_i.pt`, consumed by the greedy optimiser) and the four histograms per layer."""
    dev = model.output.weight.device
    bsz, seq_len = input_ids.shape
    fc, mask = _prefill_tables(model, seq_len, dev)
    hidden = model.tok_embeddings(input_ids.to(dev))
    if save_activations:
        os.makedirs(os.path.join(output_path, "activations"), exist_ok=True)
    for i, layer in enumerate(model.layers):
        if save_activations:
            torch.save(hidden.cpu(), os.path.join(output_path, "activations", f"act_{i}.pt"))
        rec: Dict[str, torch.Tensor] = {}
        hidden = layer_forward(layer, hidden, fc, mask, record=rec)
        for sub, keys in (("self_attn", ("attn_h1", "attn_h2")), ("mlp", ("mlp_h1", "mlp_h2"))):
            hist = {}
            for name, key in zip(("h1", "h2"), keys):
                counts, centers = find_histogram(rec[key].reshape(-1, rec[key].shape[-1]), num_bins)
                hist[name], hist[f"{name}_centers"] = counts, centers
            d = os.path.join(output_path, "histograms", f"layer-{i}", sub)
            os.makedirs(d, exist_ok=True)
            torch.save(hist, os.path.join(d, "histograms.pt"))


def effective_sparsity(sparsities: Dict[str, float], weights: Dict[str, float]) -> float:
    """parameter-weighted mean sparsity (teal/greedyopt.py:64-73)."""
    return sum(sparsities[p] * weights[p] for p in weights) / sum(weights.values())


def activation_error(target, new, last_fraction=0.25):
    start = int(new.shape[1] * (1 - last_fraction))
    return torch.norm(target[:, start:] - new[:, start:], dim=1).mean()


def _sparse_fns(hist_dir: str, layer_idx: int) -> Dict[str, SparsifyFn]:
    from .monkeypatch import PROJ_HIST
    cache, out = {}, {}
    for p in PROJS:
        sub, h = PROJ_HIST[p]
        if (sub, h) not in cache:
            cache[(sub, h)] = Distribution(os.path.join(hist_dir, f"layer-{layer_idx}", sub), h)
        out[p] = SparsifyFn(cache[(sub, h)])
    return out


@torch.no_grad()
def greedy_optimize_layer(layer: TransformerBlock, layer_idx: int, input_acts: torch.Tensor, hist_dir: str,
                          out_csv: str, weights: Dict[str, float], freqs_cis, mask, target_sparsity: float = 0.9,
                          base_step_size: float = 0.05, last_fraction: float = 0.25) -> Dict[str, float]:
    """teal/greedyopt.py:99-159: repeatedly raise the sparsity of the projection whose increment hurts the
    block output least (L2 error on the last quarter of the sequence), one CSV row per step, until the
    parameter-weighted sparsity reaches the target."""
    sp = _sparse_fns(hist_dir, layer_idx)

    def set_all(s):
        for p in PROJS:
            sp[p].set_threshold(s[p])

    sparsities = {p: 0.0 for p in PROJS}
    set_all(sparsities)
    target = layer_forward(layer, input_acts, freqs_cis, mask, sp)
    steps = {p: base_step_size * (1 / weights[p]) for p in PROJS}
    os.makedirs(os.path.dirname(out_csv), exist_ok=True)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Effective Sparsity", "Activation Error", "Baseline Error"] + list(PROJS))
        while effective_sparsity(sparsities, weights) < target_sparsity:
            best_err, best = float("inf"), None
            for p in PROJS:
                if sparsities[p] >= 1:
                    continue
                trial = deepcopy(sparsities)
                trial[p] += steps[p]
                set_all(trial)
                err = activation_error(target, layer_forward(layer, input_acts, freqs_cis, mask, sp), last_fraction)
                if err < best_err:
                    best_err, best = err, p
            if best is None:
                break
            sparsities[best] += steps[best]
            eff = effective_sparsity(sparsities, weights)
            set_all({p: eff for p in PROJS})
            base_err = activation_error(target, layer_forward(layer, input_acts, freqs_cis, mask, sp), last_fraction)
            w.writerow([eff, float(best_err), float(base_err)] + [sparsities[p] for p in PROJS])
    return sparsities


@torch.no_grad()
def greedy_optimize(model: Transformer, teal_path: str, weights: Optional[Dict[str, float]] = None, **kw) -> None:
    weights = weights or weights_from_config(model.config)
    hist_dir = os.path.join(teal_path, "histograms")
    for i, layer in enumerate(model.layers):
        acts = torch.load(os.path.join(teal_path, "activations", f"act_{i}.pt"), map_location=model.output.weight.device)
        fc, mask = _prefill_tables(model, acts.shape[1], acts.device)
        greedy_optimize_layer(layer, i, acts, hist_dir, os.path.join(teal_path, "lookup", f"layer-{i}", "results.csv"),
                              weights, fc, mask, **kw)


def main():
    ap = argparse.ArgumentParser(description="Build TEAL calibration files for a gpt-fast-shaped model")
    ap.add_argument("--synthetic", type=str, default=None, help="architecture name (random weights)")
    ap.add_argument("--checkpoint_path", type=str, default=None)
    ap.add_argument("--output_path", type=str, required=True)
    ap.add_argument("--bsz", type=int, default=4)
    ap.add_argument("--seq_len", type=int, default=256)
    ap.add_argument("--num_bins", type=int, default=10000)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--greedy", action="store_true", help="also run the block-wise greedy optimiser")
    ap.add_argument("--target_sparsity", type=float, default=0.9)
    ap.add_argument("--base_step_size", type=float, default=0.05)
    a = ap.parse_args()
    from pathlib import Path
    from .gpt_fast import generate as G
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a.precision]
    if a.synthetic:
        model = G.build_synthetic_model(a.synthetic, a.device, dt)
        g = torch.Generator().manual_seed(0)
        ids = torch.randint(0, model.config.vocab_size, (a.bsz, a.seq_len), generator=g)
    else:
        model = G.load_checkpoint_model(Path(a.checkpoint_path), a.device, dt)
        raise SystemExit("real-text calibration needs a tokenizer + dataset; pass --synthetic here (no network)")
    grab_histograms(model, ids, a.output_path, a.num_bins)
    if a.greedy:
        greedy_optimize(model, a.output_path, target_sparsity=a.target_sparsity, base_step_size=a.base_step_size)
    print(f"wrote {a.output_path}/histograms" + (f" and {a.output_path}/lookup" if a.greedy else ""))


if __name__ == "__main__":
    main()
