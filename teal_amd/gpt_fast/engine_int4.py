"""HIP decode step for int4 group-quantised blocks (second half).

The reference ships int4 as a first-class dense decode path (gpt-fast/quantize.py:483-526 WeightOnlyInt4Linear, loader
branch gpt-fast/generate.py:236-242) and lists quantised TEAL as missing (README.md:110).  Here the int4 sparse GEMV
(teal_gemv_int4.hip: |x| > tau mask, group-wise scale / zero applied once per (group, column), split-K over groups folded in
by arrival tickets) takes a plain activation vector, so the element-wise steps between two GEMVs are launches of their own
(teal_glue.hip) instead of producers inside the GEMV launch.  Per layer, all through the C ABI:

  1. resid_rmsnorm   h = resid + down;  x = RMSNorm(h) * w                     (layer 0: the embedding row)
  2. qkv             int4 sparse GEMV, three thresholds                          -> q|k|v (rounded)
  3. attn            RoPE, KV append, split-KV attention + merge                 -> y (rounded)
  4. wo              int4 sparse GEMV(tau_o)                                     -> o
  5. resid_rmsnorm   h = resid + o;  x = RMSNorm(h) * w
  6. gate, up        two int4 sparse GEMVs (tau_gate, tau_up)
  7. silu_mul        h = silu(gate) * up
  8. down            int4 sparse GEMV(tau_down)
  lm_head            resid_rmsnorm + dense 16-bit GEMV                           -> logits

10 launches per layer against 5 for 16-bit / int8 weights — the price of the plain-vector interface — but one hipGraph
replay per token with the sampler and the loop state on the device, like DecodeEngine, whose sampling / capture / decode
loop this class inherits.  Same rounding points as the module path (the ops round q|k|v, o, gate, up, down to the
activation dtype there too), so the two agree to rounding with every row kept (tests/test_int4.py).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import torch

from .. import _lib, runtime
from ..monkeypatch import to_column_major
from .engine import TEAL_IN_PLAIN, TEAL_OUT_ROUNDED, DecodeEngine, GemvIn, _out
from .model import Transformer


class Int4DecodeEngine(DecodeEngine):
    @staticmethod
    def supports(model: Transformer, need_caches: bool = True) -> Optional[str]:
        from ..quantize import int4_kernel_supports, is_int4
        cfg = model.config
        kv = cfg.n_local_heads * cfg.head_dim
        for layer in model.layers:
            at, ff = layer.attention, layer.feed_forward
            lins = (("wqkv", at.wqkv), ("wo", at.wo), ("w1", ff.w1), ("w3", ff.w3), ("w2", ff.w2))
            if not all(is_int4(l) for _, l in lins):
                return "not every projection is int4 group-quantised"
            for n, l in lins:
                if not int4_kernel_supports(l.in_features, l.out_features, l.groupsize, kv if n == "wqkv" else 0):
                    return f"{n} {l.in_features}x{l.out_features} g{l.groupsize} is outside the int4 kernel's shape contract"
        if not isinstance(model.output, torch.nn.Linear) or model.output.weight.dtype not in (torch.float16, torch.bfloat16):
            return "the int4 engine keeps lm_head in fp16 / bf16"
        if cfg.head_dim not in (64, 128) or cfg.dim > 16384 or cfg.dim != cfg.n_head * cfg.head_dim or cfg.vocab_size % 8:
            return "head_dim must be 64 or 128, dim <= 16384, vocab_size a multiple of 8"
        if not need_caches:
            return None if model.output.weight.is_cuda else "model is not on a HIP device"
        dt = model.output.weight.dtype
        if model.freqs_cis is None or model.freqs_cis.dtype != dt:
            return "caches are not set up (model.setup_caches) in the model dtype"
        for layer in model.layers:
            kc = getattr(layer.attention, "kv_cache", None)
            if kc is None or kc.k_cache.shape[0] != 1 or not kc.k_cache.is_contiguous() or not kc.v_cache.is_contiguous():
                return "KV caches must be contiguous with max_batch_size == 1"
        return None if model.output.weight.is_cuda else "model is not on a HIP device"

    def __init__(self, model: Transformer, thresholds: List[Dict[str, float]], pair: Optional[bool] = None, att_split: int = 0):
        why = Int4DecodeEngine.supports(model)
        if why is not None:
            raise ValueError(f"Int4DecodeEngine cannot run this model: {why}")
        self.L = _lib.load()
        runtime.init()
        cfg = model.config
        self.cfg, self.model = cfg, model
        dev, dt = model.output.weight.device, model.output.weight.dtype
        self.dtype, self.code, self.int8, self.int4, self.pair = dt, runtime.dtype_code(dt), False, True, False
        dim, inter, hd = cfg.dim, cfg.intermediate_size, cfg.head_dim
        kv = cfg.n_local_heads * hd
        self.dim, self.inter, self.kv, self.nqkv = dim, inter, kv, dim + 2 * kv
        to_column_major(model.output)
        e = lambda *shape, dtype=dt: torch.zeros(*shape, device=dev, dtype=dtype)  # noqa: E731
        self.resid = [e(dim), e(dim)]
        self.x_in, self.qkv, self.y_attn, self.o = e(dim), e(self.nqkv), e(dim), e(dim)
        self.gu, self.h_mlp, self.down = e(2 * inter), e(inter), e(dim)
        self.logits = e(1, 1, cfg.vocab_size)
        self.ws = runtime.new_workspace(max(dim, inter), max(self.nqkv, inter, cfg.vocab_size))
        self.rope = model.freqs_cis.contiguous()
        self.max_seq = model.max_seq_length
        # split-KV attention with the merge launch (the int4 wo GEMV takes the rounded attention output)
        if att_split:
            self.att_split = int(att_split)
        else:
            self.att_split = 4 if self.max_seq <= 1024 else (8 if self.max_seq <= 4096 else 16)
        self.att_fused_merge = False
        self.att_ws = e(cfg.n_head * self.att_split * (hd + 2), dtype=torch.float32)
        self.eps = float(cfg.norm_eps)
        self.rng_state = torch.tensor([1234, 0], dtype=torch.int64, device=dev)
        self._seed, self._calls = 1234, 0
        self.token = torch.zeros(1, dtype=torch.int32, device=dev)
        self.tok_buf = torch.zeros(1, 1, dtype=torch.int32, device=dev)
        self.pos_buf = torch.zeros(1, dtype=torch.int32, device=dev)
        self.history = torch.zeros(max(8, model.max_seq_length), dtype=torch.int32, device=dev)
        self._graph, self._graph_key = None, None
        self._build(thresholds)

    def _build(self, ths):
        self.ths = [dict(t) for t in ths]
        self.head_in = GemvIn(mode=TEAL_IN_PLAIN, x=self.x_in.data_ptr())
        m = self.model
        self.head_out = _out([(m.output.weight.data_ptr(), m.output.weight.stride(1), 0, self.cfg.vocab_size, float("-inf"),
                               self.logits.data_ptr(), None)], TEAL_OUT_ROUNDED)
        self.stages = list(range(len(m.layers)))  # (len(self.stages) is what the shared helpers use)

    def thresholds(self) -> List[Dict[str, float]]:
        return [dict(t) for t in self.ths]

    # ---- launches -----------------------------------------------------------------------------------------------------
    def _i4(self, x: torch.Tensor, lin, y: torch.Tensor, tq: float, tk: float, tv: float, kv: int):
        Z, N = lin.in_features, lin.out_features
        rc = self.L.teal_sparse_qkv_gemv_i4(x.data_ptr(), lin.weight.data_ptr(), lin.scales_and_zeros.data_ptr(), y.data_ptr(), tq, tk, tv,
                                            Z, N, N - 2 * kv, kv, lin.weight.stride(0), lin.groupsize, self.code, self.ws.data_ptr(),
                                            self.ws.numel() * 4, self._stream)
        if rc != 0:
            _lib.check(rc, "teal_sparse_qkv_gemv_i4")

    def _norm(self, resid_in, row_index, add, w, resid_out, x_out, Z):
        rc = self.L.teal_resid_rmsnorm(resid_in, row_index, add, w.data_ptr(), self.eps, resid_out, x_out.data_ptr(), Z, self.code, self._stream)
        if rc != 0:
            _lib.check(rc, "teal_resid_rmsnorm")

    def __call__(self, idx: torch.Tensor, input_pos: torch.Tensor, hook=None) -> torch.Tensor:
        assert idx.dtype == torch.int32 and input_pos.dtype == torch.int32 and idx.numel() == 1
        self._stream = runtime.stream_ptr()
        cfg, m = self.cfg, self.model
        A, B = self.resid
        cb = hook if hook else (lambda *a: None)
        for i, layer in enumerate(m.layers):
            at, ff, th = layer.attention, layer.feed_forward, self.ths[i]
            if i == 0:
                self._norm(m.tok_embeddings.weight.data_ptr(), idx.data_ptr(), None, layer.attention_norm.weight, B.data_ptr(), self.x_in, self.dim)
            else:
                self._norm(A.data_ptr(), None, self.down.data_ptr(), layer.attention_norm.weight, B.data_ptr(), self.x_in, self.dim)
            cb("before", "qkv", i)
            self._i4(self.x_in, at.wqkv, self.qkv, th["q"], th["k"], th["v"], self.kv)
            kc, vc = at.kv_cache.k_cache, at.kv_cache.v_cache
            rc = self.L.teal_decode_attention_split_ws(self.qkv.data_ptr(), None, 0, self.rope.data_ptr(), input_pos.data_ptr(), kc.data_ptr(),
                                                       vc.data_ptr(), self.y_attn.data_ptr(), None, 0.0, cfg.n_head, cfg.n_local_heads,
                                                       cfg.head_dim, self.max_seq, self.att_split, self.att_ws.data_ptr(),
                                                       self.att_ws.numel() * 4, self.code, self.ws.data_ptr(), self.ws.numel() * 4, self._stream)
            if rc != 0:
                _lib.check(rc, "teal_decode_attention_split_ws")
            cb("before", "wo", i)
            self._i4(self.y_attn, at.wo, self.o, th["o"], th["o"], th["o"], 0)
            self._norm(B.data_ptr(), None, self.o.data_ptr(), layer.ffn_norm.weight, A.data_ptr(), self.x_in, self.dim)
            cb("before", "gate_up", i)
            self._i4(self.x_in, ff.w1, self.gu[: self.inter], th["gate"], th["gate"], th["gate"], 0)
            self._i4(self.x_in, ff.w3, self.gu[self.inter:], th["up"], th["up"], th["up"], 0)
            rc = self.L.teal_silu_mul(self.gu.data_ptr(), self.gu.data_ptr() + 2 * self.inter, self.h_mlp.data_ptr(), self.inter, self.code, self._stream)
            if rc != 0:
                _lib.check(rc, "teal_silu_mul")
            cb("before", "down", i)
            self._i4(self.h_mlp, ff.w2, self.down, th["down"], th["down"], th["down"], 0)
        self._norm(A.data_ptr(), None, self.down.data_ptr(), m.norm.weight, None, self.x_in, self.dim)
        rc = self.L.teal_fused_gemv(ctypes.byref(self.head_in), ctypes.byref(self.head_out), self.dim, self.code, self.ws.data_ptr(),
                                    self.ws.numel() * 4, None, self._stream)
        if rc != 0:
            _lib.check(rc, "teal_fused_gemv")
        return self.logits

    @torch.no_grad()
    def site_activations(self, idx: torch.Tensor, input_pos: torch.Tensor) -> List[Dict[str, torch.Tensor]]:
        """|activation| every projection consumes in one decode step: every GEMV input is a plain vector here."""
        out: List[Dict[str, torch.Tensor]] = [dict() for _ in self.model.layers]
        src = {"qkv": ("attn_in", self.x_in), "wo": ("attn_out", self.y_attn), "gate_up": ("mlp_in", self.x_in), "down": ("mlp_mid", self.h_mlp)}

        def hook(when, stage, i):
            if when == "before" and i >= 0 and stage in src:
                key, buf = src[stage]
                out[i][key] = buf.float().abs().clone()

        self(idx, input_pos, hook=hook)
        return out

    @torch.no_grad()
    def calibrate_on_decode(self, sparsities, first_token, pos0, n_steps, n_samples: int = 5, rounds: int = 2):
        assert pos0 + n_steps <= self.max_seq
        ths = self.thresholds()
        for _ in range(rounds):
            pool = [dict(attn_in=[], attn_out=[], mlp_in=[], mlp_mid=[]) for _ in self.model.layers]

            def visit(acts):
                for i, a in enumerate(acts):
                    for k, v in a.items():
                        pool[i][k].append(v)

            self._walk(first_token, pos0, n_steps, n_samples, visit)
            for i in range(len(self.model.layers)):
                for p, site in self.SITE.items():
                    s = float(sparsities[p][i])
                    ths[i][p] = -1.0 if s <= 0 else float(torch.quantile(torch.cat(pool[i][site]), s))
            self._build(ths)
            self._graph = None
        return ths


def pick_engine(model: Transformer, need_caches: bool = True):
    """(engine class, None) for the fused decode step that can run `model` as it stands, or (None, reason)."""
    from ..quantize import is_int4
    if any(is_int4(l) for layer in model.layers for l in (layer.attention.wqkv, layer.feed_forward.w1)):
        why = Int4DecodeEngine.supports(model, need_caches)
        return (Int4DecodeEngine, None) if why is None else (None, why)
    why = DecodeEngine.supports(model, need_caches)
    return (DecodeEngine, None) if why is None else (None, why)
