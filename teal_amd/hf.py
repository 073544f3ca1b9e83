"""HF-transformers plugin surface (north_star "SparsifiedLinear / teal.mlp / teal.self_attn
module swap"): the reference's accuracy path masks the inputs of the 7 projections of every decoder
layer with `SparsifyFn`s kept in `layer.mlp.sparse_fns[gate|up|down]` / `layer.self_attn.sparse_fns[q|k|v|o]`
(teal/mlp.py:14-57, teal/self_attn.py:21-156) and is driven through `SparseModelMixin`
(teal/model.py:43-152).  The reference re-implements the HF forwards to insert the masks; here each
projection is wrapped instead, which gives the same maths (`proj(sparsify(x))`) for any attention
implementation and lets the S == 1 case run the HIP sparse GEMV:

    SparsifiedLinear(linear, sparse_fn)     prefill/batched: linear(sparse_fn(x))  (reference semantics,
                                            incl. the "last half of the sequence" prefill rule)
                                            decode [1, 1, Z] on the GPU: teal_sparse_gemv with the same
                                            threshold (the kernel's fp32 compare; differs from
                                            SparsifyFn's dtype-rounded compare only on boundary values)
    sparsify_hf_model(model, histogram_path)   installs sparse_fns + wrappers on every decoder layer
    SparseModelControl                          set_uniform_sparsity / set_mlp_sparsity / set_self_attn_sparsity /
                                                set_sparsities / load_greedy_sparsities / reset_sparsities /
                                                set_apply_prefill  (names of teal/model.py:96-152)
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .distribution import Distribution
from .utils import SparsifyFn, get_layer_greedy_sparsities

ATTN_PROJS = {"q": "q_proj", "k": "k_proj", "v": "v_proj", "o": "o_proj"}
MLP_PROJS = {"gate": "gate_proj", "up": "up_proj", "down": "down_proj"}


class SparsifiedLinear(nn.Module):
    """A Linear whose input is magnitude-thresholded; single-token calls read only the surviving weight
    columns through the HIP kernel."""

    def __init__(self, linear: nn.Linear, sparse_fn: SparsifyFn, use_kernel: bool = True):
        super().__init__()
        self.linear, self.sparse_fn, self.use_kernel = linear, sparse_fn, use_kernel

    @property
    def weight(self):
        return self.linear.weight

    @property
    def bias(self):
        return self.linear.bias

    def _kernel_weight(self) -> torch.Tensor:
        """the linear's OWN weight, re-laid out in place as W^T [Z][N + pad] behind the same [N, Z] view
        (monkeypatch.to_column_major: F.linear accepts the strided view, so prefill and decode read one copy and a
        load_state_dict / in-place update can never leave a stale duplicate).  A weight tensor replaced wholesale
        (load_state_dict(assign=True)) is row-major again and is simply re-laid out at the next single-token call."""
        from .monkeypatch import to_column_major
        to_column_major(self.linear)
        return self.linear.weight

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        w = self.linear.weight
        single = x.numel() == x.shape[-1]
        if (self.use_kernel and single and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and w.dtype == x.dtype
                and self.linear.bias is None and w.shape[0] % 8 == 0 and not torch.is_grad_enabled()):
            from .kernels.sparse_gemv import splitk_sparse_gemv
            y = splitk_sparse_gemv(x.reshape(1, 1, -1), self._kernel_weight(), float(self.sparse_fn.threshold), 0)
            return y.reshape(*x.shape[:-1], w.shape[0])
        xs = self.sparse_fn(x if x.dim() == 3 else x.reshape(1, -1, x.shape[-1])).reshape(x.shape)
        return F.linear(xs, w, self.linear.bias)


def _decoder_layers(model):
    inner = getattr(model, "model", model)
    return inner.layers


def sparsify_hf_model(model, histogram_path: str, apply_prefill: bool = True, use_kernel: bool = True):
    """Install the TEAL plugin on every decoder layer of an HF Llama/Mistral-style model (duck-typed on
    `self_attn.{q,k,v,o}_proj` and `mlp.{gate,up,down}_proj`).  Histogram layout and keys as in the
    reference: `<histogram_path>/layer-i/{self_attn,mlp}/histograms.pt`, h1 -> q/k/v and gate/up, h2 -> o, down."""
    for p in model.parameters():
        p.requires_grad = False
    for i, layer in enumerate(_decoder_layers(model)):
        for mod, sub, projs, h2_keys in ((layer.self_attn, "self_attn", ATTN_PROJS, ("o",)), (layer.mlp, "mlp", MLP_PROJS, ("down",))):
            path = os.path.join(histogram_path, f"layer-{i}", sub)
            mod.distrs = {"h1": Distribution(path, "h1"), "h2": Distribution(path, "h2")}
            mod.sparse_fns = nn.ModuleDict({k: SparsifyFn(mod.distrs["h2" if k in h2_keys else "h1"], apply_prefill=apply_prefill)
                                            for k in projs})
            for k, attr in projs.items():
                lin = getattr(mod, attr)
                if isinstance(lin, SparsifiedLinear):
                    lin = lin.linear
                setattr(mod, attr, SparsifiedLinear(lin, mod.sparse_fns[k], use_kernel))
    return SparseModelControl(model)


class SparseModelControl:
    """The sparsity knobs of teal/model.py:96-152 for a model prepared by sparsify_hf_model()."""

    def __init__(self, model):
        self.model = model

    def _layers(self):
        return _decoder_layers(self.model)

    def set_apply_prefill(self, apply_prefill: bool):
        for layer in self._layers():
            for fn in list(layer.self_attn.sparse_fns.values()) + list(layer.mlp.sparse_fns.values()):
                fn.apply_prefill = apply_prefill

    def set_mlp_sparsity(self, sparsity: float):
        for layer in self._layers():
            for k in MLP_PROJS:
                layer.mlp.sparse_fns[k].set_threshold(sparsity)

    def set_self_attn_sparsity(self, sparsity: float):
        for layer in self._layers():
            for k in ATTN_PROJS:
                layer.self_attn.sparse_fns[k].set_threshold(sparsity)

    def set_uniform_sparsity(self, sparsity: float):
        self.set_mlp_sparsity(sparsity)
        self.set_self_attn_sparsity(sparsity)

    def reset_sparsities(self):
        self.set_uniform_sparsity(0)

    def set_sparsities(self, sparsities: Dict[str, Sequence[float]]):
        for proj, vals in sparsities.items():
            for layer, s in zip(self._layers(), vals):
                (layer.self_attn if proj in ATTN_PROJS else layer.mlp).sparse_fns[proj].set_threshold(s)

    def load_greedy_sparsities(self, greedy_sparsity_path: str, greedy_sparsity_level: float):
        n = len(self._layers())
        self.set_sparsities(get_layer_greedy_sparsities([greedy_sparsity_level] * n, greedy_sparsity_path))
