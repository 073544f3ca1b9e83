"""Tensor parallelism for the TEAL decode path — host side (second half).

What the reference does (gpt-fast/tp.py:110-140, launched by gpt-fast/scripts/tp_run.sh): per block `wqkv`, `w1`, `w3` are
split column-wise (output features: whole heads / whole intermediate columns per rank), `wo` and `w2` row-wise (input
features), and the block's two partial outputs — attention and MLP — are summed over the ranks with ONE all-reduce of
`[B, S, dim]` each (tp.py:120-121, 139-140).  Under TEAL this composes with the activation sparsity without any new rule:

* column-wise projections (q|k|v, gate, up) see the REPLICATED activation, so every rank builds the same keep mask from the
  same threshold and streams the kept rows of ITS columns: the per-rank launch is the same sparse GEMV on a narrower image;
* row-wise projections (o, down) see the rank's SLICE of the activation (its heads' attention output, its intermediate
  columns): `|x_local| > tau` with the unchanged threshold IS the rank-local slice of the global mask, and the rank's fp32
  partial sum over its kept rows is one term of the all-reduce.

Single-batch decode is the north-star path and does not shard (BASELINE.json: "no RCCL"); TP is the capacity feature that
puts Llama-2-70B-class models on several GPUs.  Per layer and token the data path then pays two all-reduces of `dim`
16-bit values (8 KB at Llama-2-7B, 16 KB at 70B) over RCCL / xGMI — latency-bound at that size, see DESIGN.md §6 — next to
the five launches of the layer.  This module is the partition math and the collective wiring; it is exercised on CPU with
gloo at world_size 2 (tests/test_tp.py) and has NOT been timed on a multi-GPU node (the round's lease is one GPU).

Entry points mirror the reference's: `maybe_init_dist()`, `apply_tp(model)`.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import nn

from .model import Attention, FeedForward, Transformer


def _rank_world(rank: Optional[int], world: Optional[int]) -> Tuple[int, int]:
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else int(os.environ.get("LOCAL_RANK", "0"))
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    return rank, world


def maybe_init_dist() -> Optional[int]:
    """One process per GPU, launched by torch.distributed.run (gpt-fast/tp.py:37-52): returns the rank, or None when the
    job has fewer than two ranks (TP is a no-op).  RCCL ("nccl") when a GPU is visible, gloo otherwise (the CPU tests)."""
    world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if world < 2:
        return None
    rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if torch.cuda.is_available():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of rank's equal share of n (n % world == 0, like the reference's assert)"""
    if n % world:
        raise ValueError(f"{n} features do not split evenly over {world} ranks")
    step = n // world
    return rank * step, (rank + 1) * step


def shard_features(n: int, rank: int, world: int, splits: Sequence[int] = ()) -> List[Tuple[int, int]]:
    """Feature ranges a rank owns.  `splits` (q | k | v widths of the fused wqkv): the rank's share of EACH part, in order —
    its query heads, its key heads, its value heads — so that the local wqkv is again q | k | v (tp.py:69-75)."""
    if not splits:
        return [shard_range(n, rank, world)]
    if sum(splits) != n:
        raise ValueError(f"splits {list(splits)} do not add up to {n}")
    out, base = [], 0
    for w in splits:
        lo, hi = shard_range(w, rank, world)
        out.append((base + lo, base + hi))
        base += w
    return out


def _take(t: torch.Tensor, ranges: List[Tuple[int, int]], dim: int) -> torch.Tensor:
    return torch.cat([t.narrow(dim, lo, hi - lo) for lo, hi in ranges], dim=dim).contiguous()


def shard_linear(linear: nn.Module, style: str, rank: int, world: int, splits: Sequence[int] = ()) -> None:
    """In place: keep the rank's share of a linear's weight [out_features, in_features] (tp.py:55-106).
    colwise = output features (and their int8 scales), rowwise = input features.  16-bit and int8 weight-only linears;
    int4 images shard by whole 128-column tiles / 32-row groups through the same ranges on their own layout, not wired here."""
    if style not in ("colwise", "rowwise"):
        raise ValueError(style)
    if hasattr(linear, "scales_and_zeros"):
        raise NotImplementedError("int4 group-quantised linears are sharded before packing (quantise the sharded model)")
    w = linear.weight
    if style == "colwise":
        ranges = shard_features(linear.out_features, rank, world, splits)
        new_w = _take(w.data, ranges, 0)
        if hasattr(linear, "scales"):
            linear.scales = _take(linear.scales, ranges, 0)
        linear.out_features = new_w.shape[0]
    else:
        ranges = shard_features(linear.in_features, rank, world, splits)
        new_w = _take(w.data, ranges, 1)
        linear.in_features = new_w.shape[1]
    if isinstance(w, nn.Parameter):
        linear.weight = nn.Parameter(new_w, requires_grad=False)
    else:  # int8 weight-only linears keep their image in a buffer
        linear.weight = new_w


def _reduce_hook(group):
    def hook(_module, _inputs, output):
        # the block's partial output: ONE sum over the ranks (tp.py:120-121, 139-140).  16-bit partials are summed as they
        # are, like the reference's funcol.all_reduce(output, "sum")
        dist.all_reduce(output, op=dist.ReduceOp.SUM, group=group)
        return output
    return hook


def apply_tp_ffn(mlp: FeedForward, rank: int, world: int, group=None) -> None:
    shard_linear(mlp.w1, "colwise", rank, world)
    shard_linear(mlp.w3, "colwise", rank, world)
    shard_linear(mlp.w2, "rowwise", rank, world)
    mlp.register_forward_hook(_reduce_hook(group))


def apply_tp_attn(attn: Attention, rank: int, world: int, group=None) -> None:
    kv = attn.n_local_heads * attn.head_dim
    if attn.n_head % world or attn.n_local_heads % world:
        raise ValueError(f"{attn.n_head} query / {attn.n_local_heads} KV heads do not split over {world} ranks")
    shard_linear(attn.wqkv, "colwise", rank, world, [attn.dim, kv, kv])
    shard_linear(attn.wo, "rowwise", rank, world)
    # the module now owns n_head / world query heads and n_local_heads / world KV heads (tp.py:131-136)
    attn.n_head //= world
    attn.n_local_heads //= world
    attn.dim = attn.n_head * attn.head_dim
    attn.register_forward_hook(_reduce_hook(group))


def apply_tp(model: Transformer, rank: Optional[int] = None, world: Optional[int] = None, group=None) -> None:
    """Shard every block of `model` for this rank (tp.py:152-158).  Call before `setup_caches` (the KV caches are
    allocated for the rank's KV heads) and before `monkeypatch_layer` (which lays the LOCAL weight images out and takes
    the thresholds from the histograms: thresholds are properties of the activation sites, unchanged by the sharding).
    The fused single-GPU decode step (engine.py) does not span ranks: a TP model decodes through the module path."""
    rank, world = _rank_world(rank, world)
    if world < 2:
        return
    cfg = model.config
    if cfg.n_head % world or cfg.n_local_heads % world or cfg.intermediate_size % world:
        raise ValueError(f"model does not split over {world} ranks: heads {cfg.n_head}/{cfg.n_local_heads}, "
                         f"intermediate {cfg.intermediate_size}")
    for block in model.layers:
        apply_tp_ffn(block.feed_forward, rank, world, group)
        apply_tp_attn(block.attention, rank, world, group)
    # what setup_caches reads: KV heads of this rank (the residual stream, embeddings, norms and lm_head stay replicated)
    cfg.n_head //= world
    cfg.n_local_heads //= world
    model.fused_decode = False
    model.tp_world, model.tp_rank = world, rank


def collectives_per_token(model: Transformer, bytes_per_elem: int = 2) -> dict:
    """The data-path collectives of one decode token under TP: two all-reduces of [1, 1, dim] per layer."""
    dim = model.tok_embeddings.weight.shape[1]
    n = 2 * len(model.layers)
    return {"all_reduce_calls": n, "elements_each": dim, "bytes_each": dim * bytes_per_elem, "bytes_per_token": n * dim * bytes_per_elem}
