"""A gpt-fast-shaped Llama decoder: the host of the sparse-GEMV plugin.

Module / parameter names follow the reference's gpt-fast/model.py so that its checkpoints
(`tok_embeddings`, `layers.N.attention.{wqkv,wo}`, `layers.N.feed_forward.{w1,w2,w3}`,
`attention_norm`, `ffn_norm`, `norm`, `output`) load unchanged and `monkeypatch_layer`
(gpt-fast/generate.py:266-323) finds the attributes it patches:

  Attention.apply_monkeypatch / FeedForward.apply_monkeypatch   <- model.py:222-224, 270-272
  sparse attention forward (gemv1 = fused qkv, gemv2 = wo)      <- model.py:163-190
  sparse FFN forward (gemv1 on w1 and w3, gemv2 on w2)          <- model.py:258-259

Everything that is not one of the 5 projections per layer (norms, RoPE, KV cache, SDPA) is plain
PyTorch here; the fused HIP decode step lives in engine.py.  The model maths (RMSNorm in fp32,
interleaved-pair RoPE, GQA by head repetition, causal mask over a static KV cache) is the
standard Llama definition the reference also implements.
"""
from __future__ import annotations

import types
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor


def find_multiple(n: int, k: int) -> int:
    return n if n % k == 0 else n + k - n % k


# name -> architecture (gpt-fast/model.py:66-79); matched case-insensitively as a substring of
# the checkpoint directory name, longest match wins.
transformer_configs = {
    "CodeLlama-7b-Python-hf": dict(block_size=16384, vocab_size=32000, n_layer=32, dim=4096, rope_base=1000000),
    "7B": dict(n_layer=32, n_head=32, dim=4096),
    "13B": dict(n_layer=40, n_head=40, dim=5120),
    "30B": dict(n_layer=60, n_head=52, dim=6656),
    "34B": dict(n_layer=48, n_head=64, dim=8192, vocab_size=32000, n_local_heads=8, intermediate_size=22016, rope_base=1000000),
    "70B": dict(n_layer=80, n_head=64, dim=8192, n_local_heads=8, intermediate_size=28672),
    "Mistral-7B": dict(n_layer=32, n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336, vocab_size=32000),
    "llama-3-8b": dict(block_size=8192, n_layer=32, n_head=32, n_local_heads=8, dim=4096, intermediate_size=14336,
                       vocab_size=128256, rope_base=500000),
    "llama-3-70b": dict(block_size=8192, n_layer=80, n_head=64, n_local_heads=8, dim=8192, intermediate_size=28672,
                        vocab_size=128256, rope_base=500000),
    "stories15M": dict(n_layer=6, n_head=6, dim=288),
    "stories110M": dict(n_layer=12, n_head=12, dim=768),
    "tiny-test": dict(n_layer=2, n_head=4, n_local_heads=2, dim=256, intermediate_size=512, vocab_size=512, block_size=128),
    "tiny-gqa-test": dict(n_layer=2, n_head=8, n_local_heads=2, dim=512, intermediate_size=1024, vocab_size=512, block_size=128),
}


@dataclass
class ModelArgs:
    block_size: int = 2048
    vocab_size: int = 32000
    n_layer: int = 32
    n_head: int = 32
    dim: int = 4096
    intermediate_size: Optional[int] = None
    n_local_heads: int = -1
    head_dim: int = 64
    rope_base: float = 10000
    norm_eps: float = 1e-5

    def __post_init__(self):
        if self.n_local_heads == -1:
            self.n_local_heads = self.n_head
        if self.intermediate_size is None:
            self.intermediate_size = find_multiple(int(2 * 4 * self.dim / 3), 256)
        self.head_dim = self.dim // self.n_head

    @classmethod
    def from_name(cls, name: str) -> "ModelArgs":
        if name in transformer_configs:
            return cls(**transformer_configs[name])
        hits = sorted((k for k in transformer_configs if k.lower() in str(name).lower()), key=len, reverse=True)
        if not hits:
            raise KeyError(f"no transformer config matches {name!r}")
        if len(hits) > 1:
            assert len(hits[0]) != len(hits[1]), name
        return cls(**transformer_configs[hits[0]])


class KVCache(nn.Module):
    def __init__(self, max_batch_size, max_seq_length, n_heads, head_dim, dtype=torch.float16):
        super().__init__()
        shape = (max_batch_size, n_heads, max_seq_length, head_dim)
        self.register_buffer("k_cache", torch.zeros(shape, dtype=dtype))
        self.register_buffer("v_cache", torch.zeros(shape, dtype=dtype))

    def update(self, input_pos: Tensor, k_val: Tensor, v_val: Tensor):
        assert input_pos.shape[0] == k_val.shape[2]
        self.k_cache[:, :, input_pos] = k_val
        self.v_cache[:, :, input_pos] = v_val
        return self.k_cache, self.v_cache


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x: Tensor) -> Tensor:
        xf = x.float()
        normed = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + self.eps)
        return normed.type_as(x) * self.weight


def precompute_freqs_cis(seq_len: int, n_elem: int, base: float = 10000, dtype: torch.dtype = torch.float16) -> Tensor:
    """[seq_len, n_elem/2, 2] table of (cos, sin) of position * base^(-2i/n_elem)."""
    inv = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
    ang = torch.outer(torch.arange(seq_len, device=inv.device).float(), inv)
    return torch.stack((torch.cos(ang), torch.sin(ang)), dim=-1).to(dtype)


def apply_rotary_emb(x: Tensor, freqs_cis: Tensor) -> Tensor:
    """rotate interleaved pairs (x[2i], x[2i+1]); x: [B, S, H, D], freqs_cis: [S, D/2, 2]."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    fc = freqs_cis.view(1, xs.size(1), 1, xs.size(3), 2).float()
    re = xs[..., 0] * fc[..., 0] - xs[..., 1] * fc[..., 1]
    im = xs[..., 1] * fc[..., 0] + xs[..., 0] * fc[..., 1]
    return torch.stack((re, im), dim=-1).flatten(3).type_as(x)


class Attention(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        assert config.dim % config.n_head == 0
        total = (config.n_head + 2 * config.n_local_heads) * config.head_dim
        self.wqkv = nn.Linear(config.dim, total, bias=False)
        self.wo = nn.Linear(config.dim, config.dim, bias=False)
        self.kv_cache: Optional[KVCache] = None
        self.n_head, self.head_dim = config.n_head, config.head_dim
        self.n_local_heads, self.dim = config.n_local_heads, config.dim
        self._register_load_state_dict_pre_hook(self._fuse_qkv_hook)

    def _fuse_qkv_hook(self, state_dict, prefix, *args):
        if prefix + "wq.weight" in state_dict:  # HF-style separate q/k/v -> fused wqkv
            parts = [state_dict.pop(prefix + f"w{n}.weight") for n in "qkv"]
            state_dict[prefix + "wqkv.weight"] = torch.cat(parts)

    def apply_monkeypatch(self):
        self.old_forward = self.forward
        self.forward = types.MethodType(_sparse_attn_forward, self)

    def _attend(self, qkv: Tensor, freqs_cis: Tensor, mask: Tensor, input_pos: Optional[Tensor]) -> Tensor:
        bsz, seqlen, _ = qkv.shape
        kv_size = self.n_local_heads * self.head_dim
        q, k, v = qkv.split([self.dim, kv_size, kv_size], dim=-1)
        q = apply_rotary_emb(q.view(bsz, seqlen, self.n_head, self.head_dim), freqs_cis).transpose(1, 2)
        k = apply_rotary_emb(k.view(bsz, seqlen, self.n_local_heads, self.head_dim), freqs_cis).transpose(1, 2)
        v = v.view(bsz, seqlen, self.n_local_heads, self.head_dim).transpose(1, 2)
        if self.kv_cache is not None:
            k, v = self.kv_cache.update(input_pos, k, v)
        rep = self.n_head // self.n_local_heads
        if rep > 1:
            k, v = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0)
        return y.transpose(1, 2).contiguous().view(bsz, seqlen, self.dim)

    def forward(self, x: Tensor, freqs_cis: Tensor, mask: Tensor, input_pos: Optional[Tensor] = None) -> Tensor:
        return self.wo(self._attend(self.wqkv(x), freqs_cis, mask, input_pos))  # dense baseline


def _sparse_attn_forward(self: Attention, x: Tensor, freqs_cis: Tensor, mask: Tensor,
                         input_pos: Optional[Tensor] = None) -> Tensor:
    """gemv1 = teal::sparse_qkv_gemv, gemv2 = teal::sparse_gemv (prefill handled inside the ops)."""
    kv_size = self.n_local_heads * self.head_dim
    if getattr(self, "int4", False):  # int4 group-quantised projections: the ops take scales_and_zeros next to the packed weight
        qkv = self.gemv1(x, self.wqkv.weight, self.wqkv.scales_and_zeros, self.thresh_q, self.thresh_k, self.thresh_v,
                         self.sparsity_bin, kv_size)
        y = self._attend(qkv, freqs_cis, mask, input_pos)
        return self.gemv2(y, self.wo.weight, self.wo.scales_and_zeros, self.thresh_o, self.sparsity_bin)
    if getattr(self, "int8", False):  # int8 weight-only projections: the ops take the scales next to the weight
        qkv = self.gemv1(x, self.wqkv.weight, self.wqkv.scales, self.thresh_q, self.thresh_k, self.thresh_v,
                         self.sparsity_bin, kv_size)
        y = self._attend(qkv, freqs_cis, mask, input_pos)
        return self.gemv2(y, self.wo.weight, self.wo.scales, self.thresh_o, self.sparsity_bin)
    qkv = self.gemv1(x, self.wqkv.weight, self.thresh_q, self.thresh_k, self.thresh_v, self.sparsity_bin, kv_size)
    y = self._attend(qkv, freqs_cis, mask, input_pos)
    return self.gemv2(y, self.wo.weight, self.thresh_o, self.sparsity_bin)


class FeedForward(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        self.w1 = nn.Linear(config.dim, config.intermediate_size, bias=False)  # gate
        self.w3 = nn.Linear(config.dim, config.intermediate_size, bias=False)  # up
        self.w2 = nn.Linear(config.intermediate_size, config.dim, bias=False)  # down

    def apply_monkeypatch(self):
        self.old_forward = self.forward
        self.forward = types.MethodType(_sparse_ffn_forward, self)

    def forward(self, x: Tensor) -> Tensor:
        return self.w2(F.silu(self.w1(x)) * self.w3(x))  # dense baseline


def _sparse_ffn_forward(self: FeedForward, x: Tensor) -> Tensor:
    if getattr(self, "int4", False):
        gate = self.gemv1(x, self.w1.weight, self.w1.scales_and_zeros, self.thresh_gate, self.sparsity_bin)
        up = self.gemv1(x, self.w3.weight, self.w3.scales_and_zeros, self.thresh_up, self.sparsity_bin)
        return self.gemv2(F.silu(gate) * up, self.w2.weight, self.w2.scales_and_zeros, self.thresh_down, self.sparsity_bin)
    if getattr(self, "int8", False):
        gate = self.gemv1(x, self.w1.weight, self.w1.scales, self.thresh_gate, self.sparsity_bin)
        up = self.gemv1(x, self.w3.weight, self.w3.scales, self.thresh_up, self.sparsity_bin)
        return self.gemv2(F.silu(gate) * up, self.w2.weight, self.w2.scales, self.thresh_down, self.sparsity_bin)
    gate = self.gemv1(x, self.w1.weight, self.thresh_gate, self.sparsity_bin)
    up = self.gemv1(x, self.w3.weight, self.thresh_up, self.sparsity_bin)
    return self.gemv2(F.silu(gate) * up, self.w2.weight, self.thresh_down, self.sparsity_bin)


class TransformerBlock(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        self.attention = Attention(config)
        self.feed_forward = FeedForward(config)
        self.ffn_norm = RMSNorm(config.dim, config.norm_eps)
        self.attention_norm = RMSNorm(config.dim, config.norm_eps)

    def forward(self, x: Tensor, input_pos: Tensor, freqs_cis: Tensor, mask: Tensor) -> Tensor:
        h = x + self.attention(self.attention_norm(x), freqs_cis, mask, input_pos)
        return h + self.feed_forward(self.ffn_norm(h))


class Transformer(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        self.config = config
        self.tok_embeddings = nn.Embedding(config.vocab_size, config.dim)
        self.layers = nn.ModuleList(TransformerBlock(config) for _ in range(config.n_layer))
        self.norm = RMSNorm(config.dim, eps=config.norm_eps)
        self.output = nn.Linear(config.dim, config.vocab_size, bias=False)
        self.freqs_cis: Optional[Tensor] = None
        self.causal_mask: Optional[Tensor] = None
        self.max_batch_size = -1
        self.max_seq_length = -1

    @classmethod
    def from_name(cls, name: str) -> "Transformer":
        return cls(ModelArgs.from_name(name))

    def setup_caches(self, max_batch_size: int, max_seq_length: int):
        if self.max_seq_length >= max_seq_length and self.max_batch_size >= max_batch_size:
            return
        max_seq_length = find_multiple(max_seq_length, 8)
        self.max_seq_length, self.max_batch_size = max_seq_length, max_batch_size
        dtype, dev = self.output.weight.dtype, self.output.weight.device
        if hasattr(self.output, "scales"):  # quantised layers carry the model dtype in their scales (gpt-fast/model.py:122-126)
            dtype = self.output.scales.dtype
        c = self.config
        for b in self.layers:
            b.attention.kv_cache = KVCache(max_batch_size, max_seq_length, c.n_local_heads, c.head_dim, dtype).to(dev)
        self.freqs_cis = precompute_freqs_cis(c.block_size, c.head_dim, c.rope_base, dtype).to(dev)
        self.causal_mask = torch.tril(torch.ones(max_seq_length, max_seq_length, dtype=torch.bool, device=dev))

    # ---- fused decode: what a reference-harness user gets after monkeypatch_layer() on every block ----------------
    # The reference relies on Inductor to fuse the ~13 framework ops per layer around its sparse GEMVs
    # (gpt-fast/generate.py:420 torch.compile(decode_one_token)).  There is no tracing compiler here: a single-token
    # call on a fully patched model runs the hand-fused HIP decode step instead (teal_amd/gpt_fast/engine.py: five
    # launches per layer, same thresholds, same weights, same KV caches, same attribute bundle on the modules).
    # `fused_decode = False` keeps the op-by-op module path (tests compare the two).
    fused_decode: bool = True

    def _patched_thresholds(self):
        ths = []
        for layer in self.layers:
            at, ff = layer.attention, layer.feed_forward
            if not (hasattr(at, "thresh_q") and hasattr(at, "gemv1") and hasattr(ff, "thresh_gate") and hasattr(ff, "gemv2")):
                return None
            ths.append({"q": float(at.thresh_q), "k": float(at.thresh_k), "v": float(at.thresh_v), "o": float(at.thresh_o),
                        "gate": float(ff.thresh_gate), "up": float(ff.thresh_up), "down": float(ff.thresh_down)})
        return ths

    def _engine_key(self, ths):
        # everything the engine's launch descriptors hold raw pointers to, plus the thresholds: KV caches re-allocated by
        # setup_caches, weights replaced by load_state_dict(assign=True) / .to(), new thresholds -> rebuild
        return (self.max_seq_length, tuple(tuple(t.values()) for t in ths), self.output.weight.data_ptr(),
                self.tok_embeddings.weight.data_ptr(), self.norm.weight.data_ptr()) + tuple(
            p for layer in self.layers for p in (layer.attention.kv_cache.k_cache.data_ptr(), layer.attention.kv_cache.v_cache.data_ptr(),
                                                 layer.attention.wqkv.weight.data_ptr(), layer.attention.wo.weight.data_ptr(),
                                                 layer.feed_forward.w1.weight.data_ptr(), layer.feed_forward.w2.weight.data_ptr(),
                                                 layer.feed_forward.w3.weight.data_ptr(), layer.attention_norm.weight.data_ptr(),
                                                 layer.ffn_norm.weight.data_ptr()))

    def _fused_engine(self):
        """The fused engine for this model's current weights / caches / thresholds, or None when the model is not fully
        patched or DecodeEngine.supports() names a reason (head_dim 48, batched caches, mixed int8 / 16-bit, ...): the
        caller then runs the op-by-op module path, as before the engine existed.  The verdict is cached per key."""
        ths = self._patched_thresholds()
        if ths is None:
            return None
        key = self._engine_key(ths)
        if getattr(self, "_eng_key", None) != key:
            from .engine import pick_engine
            cls, why = pick_engine(self)  # DecodeEngine, or the reason it cannot run this model
            object.__setattr__(self, "_eng_why", why)
            # not a submodule: the engine only borrows this model
            object.__setattr__(self, "_eng", None if cls is None else cls(self, ths))
            self._eng_key = self._engine_key(ths)  # after the build: the engine re-lays lm_head out column-major
        return self._eng

    def forward(self, idx: Tensor, input_pos: Optional[Tensor] = None) -> Tensor:
        assert self.freqs_cis is not None, "Caches must be initialized first"
        if (self.fused_decode and idx.numel() == 1 and idx.is_cuda and input_pos is not None and input_pos.numel() == 1
                and not torch.is_grad_enabled() and self.output.weight.dtype in (torch.float16, torch.bfloat16, torch.int8)):
            eng = self._fused_engine()
            if eng is not None:
                # a fresh tensor like the op-by-op path returns (the engine's logits buffer is overwritten by the next step)
                return eng(idx.view(1, 1).to(torch.int32), input_pos.view(1).to(torch.int32)).clone()
        mask = self.causal_mask[None, None, input_pos]
        freqs_cis = self.freqs_cis[input_pos]
        x = self.tok_embeddings(idx)
        for layer in self.layers:
            x = layer(x, input_pos, freqs_cis, mask)
        return self.output(self.norm(x))
