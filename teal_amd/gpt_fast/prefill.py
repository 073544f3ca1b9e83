"""The prompt pass of a SHORT prompt (T <= 16 tokens) as hand-fused HIP launches — dense, like the reference's.

The reference's prefill is dense by construction: `SparseGEMV.forward` / `SparseQKVGEMV.forward` run `torch.matmul(x, W.T)`
when the sequence is longer than one token (kernels/sparse_gemv.py:271,298) inside the stock gpt-fast forward
(gpt-fast/model.py:107-121, 158-186, 258-259, 289-291), and its tokens/sec counts that pass (gpt-fast/generate.py:458,487-496).
Op by op, the 6-token default prompt costs ~400 launches here: 20 ms eager, 10 ms from a hipGraph — six to twelve decode steps'
worth (profiles/r05_generate_breakdown_before.txt).  `PrefillEngine` runs the same pass as seven launches per layer over the
decode step's own weight images (teal_amd/csrc/teal_prefill.hip):

    gemm(wqkv) [RMSNorm while staging] -> attention (RoPE, cache rows 0..T-1, causal softmax) -> gemm(wo) -> resid ->
    gemm(w1 | w3) [RMSNorm while staging] -> gemm(w2) [silu * up while staging] -> resid
                                                           ... -> norm + lm_head of the LAST token (teal_dense_gemv)

and returns logits [1, 1, vocab] of the last prompt token (all `generate()` samples from).  Nothing is sparsified: thresholds
play no role in the prompt pass.  `FusedPrefill` is what generate() calls: the HIP pass (replayed from a hipGraph per prompt
length) where it applies — 16-bit weights, one GPU, 2 <= T <= 16, positions 0..T-1 — and the module path otherwise.
Round 6: 9-16 tokens (two 16-byte words per feature in the hand-over layout) — they cost 2.1-2.2x the 8-token pass through the
module path (profiles/r06_prefill_vs_prompt_length.txt).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import _lib, runtime
from ..monkeypatch import UP_SHIFT_BYTES, to_column_major
from .model import Transformer

MAX_T = 16  # most tokens of a pass; the hand-over layout is [feature][8] for T <= 8 and [feature][16] for 9 <= T <= 16 (buffers sized for 16)
IN_XT, IN_NORM, IN_SILU_MUL = 0, 1, 2


class PrefillIn(ctypes.Structure):  # teal_prefill_in_t
    _fields_ = [("mode", ctypes.c_int), ("xt", ctypes.c_void_p), ("sumsq", ctypes.c_void_p), ("nwg", ctypes.c_int),
                ("norm_w", ctypes.c_void_p), ("eps", ctypes.c_float), ("gu_slabs", ctypes.c_void_p), ("gu_split", ctypes.c_int)]


class PrefillEngine:
    @staticmethod
    def supports(model: Transformer) -> Optional[str]:
        """None if the fused prompt pass can run `model` as it stands, else the reason (the caller keeps the module path)."""
        cfg = model.config
        if int(getattr(model, "tp_world", 1)) > 1:
            return "tensor-parallel models prefill through the module path (its all-reduce hooks)"
        lins = [lin for layer in model.layers for lin in (layer.attention.wqkv, layer.attention.wo, layer.feed_forward.w1,
                                                         layer.feed_forward.w3, layer.feed_forward.w2)] + [model.output]
        if any(hasattr(lin, "scales_and_zeros") or hasattr(lin, "scales") for lin in lins):
            return "quantised weights prefill through the module path"
        dt = model.output.weight.dtype
        if dt not in (torch.float16, torch.bfloat16) or any(lin.weight.dtype != dt for lin in lins):
            return f"weights are not uniformly fp16 / bf16: {dt}"
        if not model.output.weight.is_cuda:
            return "model is not on a HIP device"
        inter = model.layers[0].feed_forward.w1.out_features
        qd, kv = cfg.n_head * cfg.head_dim, cfg.n_local_heads * cfg.head_dim
        if cfg.head_dim not in (64, 128) or cfg.dim != qd or cfg.dim % 256 or cfg.dim > 16384 or inter % 256 or (qd + 2 * kv) % 256:
            return "shape outside the prompt-pass kernels' contract (head_dim 64 / 128, dim = n_head * head_dim <= 16384, widths % 256)"
        if cfg.vocab_size % 8:
            return "vocab_size must be a multiple of 8"
        if model.freqs_cis is None or model.freqs_cis.dtype != dt:
            return "caches are not set up (model.setup_caches) in the model dtype"
        for layer in model.layers:
            kc = getattr(layer.attention, "kv_cache", None)
            if kc is None or kc.k_cache.shape[0] != 1 or not kc.k_cache.is_contiguous() or not kc.v_cache.is_contiguous():
                return "KV caches must be contiguous with max_batch_size == 1"
        return None

    def __init__(self, model: Transformer):
        why = PrefillEngine.supports(model)
        if why is not None:
            raise ValueError(f"PrefillEngine cannot run this model: {why}")
        self.L = _lib.load()
        runtime.init()
        self.model, cfg = model, model.config
        dev, dt = model.output.weight.device, model.output.weight.dtype
        self.code = runtime.dtype_code(dt)
        for layer in model.layers:  # the decode step's layout (idempotent: monkeypatch_layer / DecodeEngine did it already)
            for lin in (layer.attention.wqkv, layer.attention.wo, layer.feed_forward.w1, layer.feed_forward.w3, layer.feed_forward.w2):
                to_column_major(lin, shift_bytes=UP_SHIFT_BYTES if lin is layer.feed_forward.w3 else 0)
        to_column_major(model.output)
        self.dim, self.hd = cfg.dim, cfg.head_dim
        self.kv = cfg.n_local_heads * cfg.head_dim
        self.nqkv = self.dim + 2 * self.kv
        self.inter = model.layers[0].feed_forward.w1.out_features
        self.max_seq = model.max_seq_length
        z = lambda *shape, dtype=dt: torch.zeros(*shape, device=dev, dtype=dtype)  # noqa: E731
        self.tokens = z(MAX_T, dtype=torch.int32)
        self.ht, self.yt = z(self.dim, MAX_T), z(self.dim, MAX_T)
        # two slab buffers [slices <= 16][columns][8 or 16] fp32, used alternately (launches are stream-ordered: a consumer has read its
        # producer's slabs before the next-but-one GEMM overwrites them; the down projection reads gate | up's while writing its own)
        n = 16 * max(self.nqkv, 2 * self.inter, self.dim) * MAX_T
        self.slabs = [z(n, dtype=torch.float32), z(n, dtype=torch.float32)]
        self.x_last = z(self.dim)
        self.sumsq = z(64 * MAX_T, dtype=torch.float32)  # per-workgroup sums of squares: resid launch -> normalising consumer
        self.logits = z(1, 1, cfg.vocab_size)
        self.rope = model.freqs_cis.contiguous()
        self.ws = runtime.new_workspace(self.dim, cfg.vocab_size)
        self.eps = float(cfg.norm_eps)
        self.nwg = (self.dim + 255) // 256
        self._split = ctypes.c_int(0)

    def key(self):
        """everything the launches hold raw pointers to: a re-laid-out weight or a re-allocated KV cache needs a new engine"""
        m = self.model
        return (m.max_seq_length, m.output.weight.data_ptr(), m.tok_embeddings.weight.data_ptr(), m.freqs_cis.data_ptr()) + tuple(
            p for layer in m.layers for p in (layer.attention.kv_cache.k_cache.data_ptr(), layer.attention.kv_cache.v_cache.data_ptr(),
                                              layer.attention.wqkv.weight.data_ptr(), layer.attention.wo.weight.data_ptr(),
                                              layer.feed_forward.w1.weight.data_ptr(), layer.feed_forward.w2.weight.data_ptr(),
                                              layer.feed_forward.w3.weight.data_ptr()))

    def _gemm(self, gin: PrefillIn, lin0, lin1, Z, T, out: torch.Tensor, st) -> int:
        w0 = lin0.weight
        n1 = lin1.weight.shape[0] if lin1 is not None else 0
        rc = self.L.teal_prefill_gemm(ctypes.byref(gin), w0.data_ptr(), w0.stride(1), w0.shape[0],
                                      lin1.weight.data_ptr() if lin1 is not None else None, lin1.weight.stride(1) if lin1 is not None else 0, n1,
                                      out.data_ptr(), out.numel() * 4, Z, T, self.code, ctypes.byref(self._split), st)
        if rc != 0:
            _lib.check(rc, "teal_prefill_gemm")
        return self._split.value

    def _resid(self, tokens: bool, slabs: Optional[torch.Tensor], split: int, T, st, final_norm=None):
        """h (+)= round(sum slabs); leaves the per-workgroup sums of squares for the normalising consumer.  final_norm: also
        the normalised vector of the LAST token as a plain vector (the lm_head's input)."""
        rc = self.L.teal_prefill_resid_norm(self.model.tok_embeddings.weight.data_ptr() if tokens else None,
                                            self.tokens.data_ptr() if tokens else None, T, None if tokens else self.ht.data_ptr(),
                                            slabs.data_ptr() if slabs is not None else None, split,
                                            final_norm.data_ptr() if final_norm is not None else None, self.eps, self.dim, self.ht.data_ptr(),
                                            None, self.x_last.data_ptr() if final_norm is not None else None, self.sumsq.data_ptr(), self.code, st)
        if rc != 0:
            _lib.check(rc, "teal_prefill_resid_norm")

    def _norm_in(self, norm_w) -> PrefillIn:
        return PrefillIn(mode=IN_NORM, xt=self.ht.data_ptr(), sumsq=self.sumsq.data_ptr(), nwg=self.nwg, norm_w=norm_w.data_ptr(), eps=self.eps)

    @torch.no_grad()
    def __call__(self, prompt: torch.Tensor) -> torch.Tensor:
        """prompt: int tokens [T], 1 <= T <= 16, occupying positions 0 .. T-1 -> logits [1, 1, vocab] of the last token; the KV
        rows 0 .. T-1 of every layer are written."""
        T = int(prompt.numel())
        assert 1 <= T <= MAX_T and T <= self.max_seq
        m, cfg, L, st = self.model, self.model.config, self.L, runtime.stream_ptr()
        self.tokens[:T].copy_(prompt.view(-1))
        layers = list(m.layers)
        A, B = self.slabs
        self._resid(True, None, 0, T, st)
        for i, layer in enumerate(layers):
            at, ff = layer.attention, layer.feed_forward
            ns = self._gemm(self._norm_in(layer.attention_norm.weight), at.wqkv, None, self.dim, T, A, st)
            kc, vc = at.kv_cache.k_cache, at.kv_cache.v_cache
            rc = L.teal_prefill_attention(A.data_ptr(), ns, self.rope.data_ptr(), kc.data_ptr(), vc.data_ptr(), self.yt.data_ptr(), T,
                                          cfg.n_head, cfg.n_local_heads, cfg.head_dim, self.max_seq, self.code, st)
            if rc != 0:
                _lib.check(rc, "teal_prefill_attention")
            ns = self._gemm(PrefillIn(mode=IN_XT, xt=self.yt.data_ptr()), at.wo, None, self.dim, T, B, st)
            self._resid(False, B, ns, T, st)
            ns = self._gemm(self._norm_in(layer.ffn_norm.weight), ff.w1, ff.w3, self.dim, T, A, st)
            ns = self._gemm(PrefillIn(mode=IN_SILU_MUL, gu_slabs=A.data_ptr(), gu_split=ns), ff.w2, None, self.inter, T, B, st)
            self._resid(False, B, ns, T, st, final_norm=m.norm.weight if i + 1 == len(layers) else None)
        w = m.output.weight
        rc = L.teal_sparse_qkv_gemv_ld(self.x_last.data_ptr(), w.data_ptr(), w.stride(1), self.logits.data_ptr(), float("-inf"), float("-inf"),
                                       float("-inf"), self.dim, cfg.vocab_size, cfg.vocab_size, 0, self.code, self.ws.data_ptr(),
                                       self.ws.numel() * 4, st)
        if rc != 0:
            _lib.check(rc, "teal_dense_gemv (lm_head)")
        return self.logits


class FusedPrefill:
    """generate()'s `prefill` callable: the HIP prompt pass (one hipGraph per prompt length when `graph`) for prompts it covers,
    `fallback(prompt)` (GraphedPrefill, or None = the eager module path) for everything else."""

    def __init__(self, model: Transformer, graph: bool = True, fallback=None):
        self.model, self.graph, self.fallback = model, graph, fallback
        self._eng: Optional[PrefillEngine] = None
        self._eng_key = None
        self._why: Optional[str] = None
        self._graphs = {}
        self.used = None  # "hip" / "fallback": what the last call ran (tests, reports)

    def _engine(self) -> Optional[PrefillEngine]:
        if self._eng is not None and self._eng.key() == self._eng_key:
            return self._eng
        self._why = PrefillEngine.supports(self.model)
        self._eng, self._graphs = (None if self._why is not None else PrefillEngine(self.model)), {}
        self._eng_key = self._eng.key() if self._eng is not None else None
        return self._eng

    def __call__(self, prompt: torch.Tensor) -> torch.Tensor:
        T = int(prompt.numel())
        # (a ONE-token prompt is a decode step in the reference too — its ops take the sparse kernel whenever the sequence length
        #  is 1, kernels/sparse_gemv.py:271,298 — so it goes the way single-token calls go: the fallback / the model's fused step)
        eng = self._engine() if 2 <= T <= MAX_T else None
        if eng is None:
            self.used = "fallback"
            if self.fallback is not None:
                return self.fallback(prompt)
            return self.model(prompt.view(1, -1), torch.arange(0, T, device=prompt.device))
        self.used = "hip"
        if not self.graph:
            return eng(prompt)
        if T not in self._graphs:
            static = prompt.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                eng(static)  # warm-up outside capture
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with runtime.graph_capture(g):
                logits = eng(static)
            self._graphs[T] = (g, static, logits)
        g, static, logits = self._graphs[T]
        static.copy_(prompt)
        g.replay()
        return logits
