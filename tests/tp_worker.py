"""Worker of tests/test_tp.py (world_size 2, gloo, CPU): the TP partition math and collective wiring of
teal_amd/gpt_fast/tp.py (reference: gpt-fast/tp.py:110-140)."""
import copy
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import teal_oracle as O  # noqa: E402  (test infrastructure: the checker)
from teal_amd.gpt_fast import tp  # noqa: E402
from teal_amd.gpt_fast.model import ModelArgs, Transformer  # noqa: E402


def bits16(t):
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16).reshape(-1)


def main():
    rank = tp.maybe_init_dist()
    assert rank is not None
    world = dist.get_world_size()
    res = {"world": world}

    # ---- 1. a sharded model reproduces the unsharded one (dense module path, fp32, CPU) -----------------------------
    cfg = ModelArgs(block_size=64, vocab_size=96, n_layer=2, n_head=8, dim=256, intermediate_size=512, n_local_heads=4)
    torch.manual_seed(5)  # the same weights on every rank
    full = Transformer(cfg).float().eval()
    for p in full.parameters():
        torch.nn.init.normal_(p, std=0.05)
    sharded = copy.deepcopy(full)
    tp.apply_tp(sharded, rank, world)
    at = sharded.layers[0].attention
    res["local_shapes"] = [list(at.wqkv.weight.shape), list(at.wo.weight.shape), list(sharded.layers[0].feed_forward.w1.weight.shape),
                           list(sharded.layers[0].feed_forward.w2.weight.shape), at.n_head, at.n_local_heads]
    idx = torch.tensor([[3, 17, 42, 9, 77, 1]])
    pos = torch.arange(6)
    with torch.no_grad():
        full.setup_caches(1, 16)
        sharded.setup_caches(1, 16)
        a = full(idx, pos)
        b = sharded(idx, pos)
        res["kv_cache_heads"] = sharded.layers[0].attention.kv_cache.k_cache.shape[1]
        res["prefill_max_err"] = float((a - b).abs().max())
        res["prefill_scale"] = float(a.abs().max())
        a1 = full(torch.tensor([[5]]), torch.tensor([6]))     # one decode step on the caches
        b1 = sharded(torch.tensor([[5]]), torch.tensor([6]))
        res["decode_max_err"] = float((a1 - b1).abs().max())
    res["collectives"] = tp.collectives_per_token(sharded)
    # thresholds: averaged over the ranks for a SHARDED model only (replicas under one process group keep their own)
    mine = [{"o": 1.0 + rank, "q": 0.5}]
    res["sync_sharded"] = tp.sync_thresholds(mine, sharded)
    res["sync_replica"] = tp.sync_thresholds(mine, full)

    # ---- 2. TEAL under TP, checked with the oracle: colwise gate | up on the replicated x with the SAME threshold, rowwise
    #         down on the rank's slice of h with the SAME threshold; the all-reduced partial sums equal the unsharded truth --
    Z, I = 256, 512
    dtype = O.F16
    xb = O.hash_uniform(Z, 31, 2.0, dtype)
    w1 = O.from_bits(O.hash_uniform_c(Z * I, 32, 0.2, dtype), dtype).reshape(Z, I)   # W1^T [Z][I]
    w3 = O.from_bits(O.hash_uniform_c(Z * I, 33, 0.2, dtype), dtype).reshape(Z, I)
    w2 = O.from_bits(O.hash_uniform_c(I * Z, 34, 0.2, dtype), dtype).reshape(I, Z)   # W2^T [I][Z]
    tg, tu = 0.45, 0.55  # x = U(-1, 1): about half of the rows kept
    wb = lambda m: O.to_bits(np.ascontiguousarray(m).reshape(-1), dtype)  # noqa: E731

    def h_of(g64, u64):  # model.py:258-259 with the roundings of the 16-bit sequence
        g = O.from_bits(O.to_bits(g64.astype(np.float32), dtype), dtype).astype(np.float32)
        u = O.from_bits(O.to_bits(u64.astype(np.float32), dtype), dtype).astype(np.float32)
        s = O.from_bits(O.to_bits((g / (1.0 + np.exp(-g))).astype(np.float32), dtype), dtype)
        return O.to_bits((s * u).astype(np.float32), dtype)

    g_full = O.truth64(xb, wb(w1), Z, I, tg, dtype=dtype)
    u_full = O.truth64(xb, wb(w3), Z, I, tu, dtype=dtype)
    hb_full = h_of(g_full, u_full)
    td = float(np.median(np.abs(O.from_bits(hb_full, dtype))))  # the same on every rank: about half of h kept
    y_full = O.truth64(hb_full, wb(w2), I, Z, td, dtype=dtype)
    (lo, hi), = tp.shard_features(I, rank, world)
    g_loc = O.truth64(xb, wb(w1[:, lo:hi]), Z, hi - lo, tg, dtype=dtype)      # same x, same threshold, this rank's columns
    u_loc = O.truth64(xb, wb(w3[:, lo:hi]), Z, hi - lo, tu, dtype=dtype)
    res["colwise_exact"] = bool(np.array_equal(g_loc, g_full[lo:hi]) and np.array_equal(u_loc, u_full[lo:hi]))
    hb_loc = h_of(g_loc, u_loc)
    res["h_slice_exact"] = bool(np.array_equal(hb_loc, hb_full[lo:hi]))
    keep_loc = O.compact(hb_loc, td, dtype)                                     # |h_local| > tau ...
    keep_full = O.compact(hb_full, td, dtype)
    res["mask_is_slice"] = bool(np.array_equal(keep_loc + lo, keep_full[(keep_full >= lo) & (keep_full < hi)]))  # ... is the global mask's slice
    part = torch.from_numpy(O.truth64(hb_loc, wb(w2[lo:hi, :]), hi - lo, Z, td, dtype=dtype))  # rank's rows of W2^T
    dist.all_reduce(part)                                                      # the MLP's ONE all-reduce (tp.py:120-121)
    res["rowwise_sum_max_err"] = float(np.abs(part.numpy() - y_full).max())
    res["rowwise_scale"] = float(np.abs(y_full).max())
    res["kept_down"] = float(len(keep_full) / I)

    # fused wqkv: the rank's q | k | v heads, each with its own threshold (kernels/sparse_gemv.py:196-237)
    nq, nkv = 256, 128
    N = nq + 2 * nkv
    wq = O.from_bits(O.hash_uniform_c(Z * N, 35, 0.2, dtype), dtype).reshape(Z, N)
    t3 = (0.4, 0.5, 0.6)
    qkv_full = O.truth64(xb, wb(wq), Z, N, t3[0], t3[1], t3[2], nq, nkv, dtype)
    rng = tp.shard_features(N, rank, world, [nq, nkv, nkv])
    cols = np.concatenate([np.arange(a_, b_) for a_, b_ in rng])
    qkv_loc = O.truth64(xb, wb(wq[:, cols]), Z, len(cols), t3[0], t3[1], t3[2], nq // world, nkv // world, dtype)
    res["qkv_exact"] = bool(np.array_equal(qkv_loc, qkv_full[cols]))
    if rank == 0:
        print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
