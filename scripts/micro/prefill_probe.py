#!/usr/bin/env python3
"""Bisect the module-path prefill at a given prompt length: every torch op of one patched 7B-width layer, synchronised."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3800
dev, dt = "cuda", torch.float16
def step(name, fn):
    y = fn(); torch.cuda.synchronize(); print("ok", name, tuple(y.shape) if hasattr(y, "shape") else "", flush=True); return y
x = torch.randn(1, S, 4096, device=dev, dtype=dt) * 0.1
for N, Z in ((12288, 4096), (4096, 4096), (11008, 4096), (4096, 11008)):
    W = (torch.randn(N, Z, device=dev, dtype=dt) * 0.02).T.contiguous().T   # the reference's column-major weight view
    xin = torch.randn(1, S, Z, device=dev, dtype=dt) * 0.1
    step(f"matmul col-major {Z}->{N}", lambda: torch.matmul(xin, W.T))
    Wr = torch.randn(N, Z, device=dev, dtype=dt) * 0.02
    step(f"matmul row-major {Z}->{N}", lambda: torch.matmul(xin, Wr.T))
from teal_amd.gpt_fast import generate as G
m = step("build", lambda: G.build_synthetic_model("7B", dev, dt, seed=1, n_layer=1))
step("apply_sparsity", lambda: G.apply_sparsity(m, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True))
m.max_seq_length = -1
step("setup_caches", lambda: m.setup_caches(1, S + 128))
pos = torch.arange(0, S, device=dev)
idx = torch.randint(0, 32000, (1, S), device=dev, dtype=torch.int)
with torch.no_grad():
    mask = m.causal_mask[None, None, pos]
    fc = m.freqs_cis[pos]
    h = step("embedding", lambda: m.tok_embeddings(idx))
    L = m.layers[0]
    a = step("attention_norm", lambda: L.attention_norm(h))
    att = L.attention
    kv = att.n_local_heads * att.head_dim
    qkv = step("wqkv op", lambda: att.gemv1(a, att.wqkv.weight, att.thresh_q, att.thresh_k, att.thresh_v, att.sparsity_bin, kv))
    y = step("attend", lambda: att._attend(qkv, fc, mask, pos))
    o = step("wo op", lambda: att.gemv2(y, att.wo.weight, att.thresh_o, att.sparsity_bin))
    h2 = h + o
    f = step("ffn", lambda: L.feed_forward(L.ffn_norm(h2)))
    step("full model", lambda: m(idx, pos))
print("done")
