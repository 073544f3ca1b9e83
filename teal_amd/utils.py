"""Host-side pieces of the reference's utils/utils.py that the decode path consumes.

  SparsifyFn                    <- utils/utils.py:9-55   (semantic spec of the mask; HF plugin surface)
  get_layer_greedy_sparsities   <- utils/utils.py:243-259 (block-wise greedy lookup reader)
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .distribution import Distribution  # noqa: F401  (re-export, as utils/utils.py does)

PROJS = ["q", "k", "v", "o", "gate", "up", "down"]


class SparsifyFn(nn.Module):
    """x -> x * (|x| > threshold), threshold from a Distribution at a sparsity level.

    Same semantics as the reference module, including its compare rule: `x.abs().gt(t)` with a
    python-float threshold rounds t to x's dtype first (so boundary values can differ from the
    GEMV kernel's fp32 rule), `set_threshold(0)` gives exactly 0.0, and on
    prefill (S > 1) only the last half of the sequence is sparsified.
    """

    def __init__(self, distr, init_sparsity=None, init_threshold=None, apply_prefill=True):
        super().__init__()
        assert init_sparsity is None or init_threshold is None, "init_sparsity and init_threshold cannot both be specified"
        if init_sparsity is not None:
            thresh = distr.icdf(0.5 + init_sparsity / 2)
        elif init_threshold is not None:
            thresh = init_threshold
        else:
            thresh = 0
        self.register_buffer("a", torch.tensor([float(thresh)]).to(torch.float16))
        self.distr = distr
        self.apply_prefill = apply_prefill
        self.threshold = float(thresh)
        self.sparsity_level = init_sparsity or 0.0

    def set_threshold(self, sparsity):
        self.threshold = self.distr.icdf(0.5 + sparsity / 2).item() if sparsity != 0.0 else 0.0
        self.sparsity_level = sparsity

    def get_threshold(self):
        return self.threshold

    def apply(self, x):
        return x.abs().gt(self.threshold) * x

    def forward(self, x):
        if x.size(1) > 1:
            if not self.apply_prefill:
                return x
            half = x.size(1) // 2
            return torch.cat((x[:, :-half, :], self.apply(x[:, -half:, :])), dim=1)
        return self.apply(x)


def get_layer_greedy_sparsities(layer_sparsities, results_dir):
    """For each layer pick the results.csv row whose `Effective Sparsity` is closest to the
    layer's target; return {proj: [sparsity per layer]}.

    results.csv header (teal/greedyopt.py:123):
      Effective Sparsity,Activation Error,Baseline Error,q,k,v,o,gate,up,down
    Parsed and ranked with pandas exactly like the reference (same float parser, same argsort
    tie order), so the tables are bit-identical to tests/golden/greedy_llama2_7b.json.
    """
    import pandas as pd

    out = {p: [0.0] * len(layer_sparsities) for p in PROJS}
    for layer, target in enumerate(layer_sparsities):
        df = pd.read_csv(os.path.join(results_dir, f"layer-{layer}", "results.csv"))
        order = (df["Effective Sparsity"] - target).abs().argsort()
        row = df.iloc[order[:1]]
        for p in PROJS:
            out[p][layer] = float(row[p].values[0])
    return out
