"""Calibration histograms -> thresholds.  Load-time, CPU, fp32; never on the per-token path.

Restates gpt-fast/distribution.py:17-66 (== utils/utils.py:72-123) of the reference: a
`histograms.pt` file is a torch-saved dict of fp32 [10000] tensors `h1`, `h1_centers`, `h2`,
`h2_centers` (bin 0 / bin 9999 are the 1 % outlier bins, utils/utils.py:145-173).  `icdf(q)`
finds the first bin whose cumulative count reaches q*total and interpolates linearly between
the neighbouring bin centres.  The arithmetic is kept in torch fp32 ops of the same kind and
order so the thresholds are bit-identical to the reference's (tests/golden/thresholds.json).
"""
from __future__ import annotations

import os

import torch

__all__ = ["Distribution", "interp", "threshold_for_sparsity"]


def interp(x: torch.Tensor, xp: torch.Tensor, fp: torch.Tensor) -> torch.Tensor:
    """piecewise-linear interpolation of (xp, fp) at x."""
    i = torch.clamp(torch.searchsorted(xp, x), 1, len(xp) - 1)
    x0, x1, f0, f1 = xp[i - 1], xp[i], fp[i - 1], fp[i]
    return f0 + (x - x0) / (x1 - x0) * (f1 - f0)


class Distribution:
    """One activation histogram (`h1` = block input, `h2` = intermediate) of one layer's
    `mlp` or `self_attn` directory."""

    def __init__(self, file_path: str, hidden_type: str):
        self.file_path = file_path
        self.hidden_type = hidden_type
        hist = torch.load(os.path.join(file_path, "histograms.pt"), map_location="cpu", weights_only=True)
        self.bin_centers = hist[f"{hidden_type}_centers"].float()
        self.counts = hist[hidden_type].float()
        self.total_count = self.counts.sum()
        self.cumulative_counts = torch.cumsum(self.counts, dim=0)

    def cdf(self, x):
        return interp(x, self.bin_centers, self.cumulative_counts / self.total_count)

    def icdf(self, q):
        """value v with P(X <= v) = q (signed histogram; assumes zero-mean unimodal)."""
        target = q * self.total_count
        idx = int(torch.searchsorted(self.cumulative_counts, target))
        n = len(self.bin_centers)
        if idx == 0:
            return self.bin_centers[0]
        if idx == n:
            return self.bin_centers[-1]
        c_lo, c_hi = self.cumulative_counts[idx - 1], self.cumulative_counts[idx]
        v_lo, v_hi = self.bin_centers[idx - 1], self.bin_centers[idx]
        frac = (target - c_lo) / (c_hi - c_lo)
        return v_lo + frac * (v_hi - v_lo)


def threshold_for_sparsity(distr: Distribution, sparsity: float) -> float:
    """tau such that P(|x| <= tau) = sparsity for a symmetric distribution:
    icdf(0.5 + 0.5*s)   (gpt-fast/generate.py:277-287)."""
    return distr.icdf(0.5 + 0.5 * sparsity).item()
