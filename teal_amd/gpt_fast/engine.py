"""Fused HIP decode step for a gpt-fast-shaped Llama.

The reference runs one token through ~13 framework ops per layer around its 5 sparse GEMVs
(gpt-fast/model.py:158-190, 258-259, 289-291) and relies on Inductor to fuse them.  Here the whole
layer is 5 hand-written HIP launches, with everything between two GEMVs folded into the consumer's
prologue (every workgroup recomputes the tiny vector work from L2; nothing round-trips through a
separate kernel):

  1. qkv     [h = resid + round(sum down-slabs); x = RMSNorm(h) * w] -> mask(tau_q|k|v) -> GEMV -> q|k|v
             (no split-K, 16-bit weights: RoPE(q, k) and the KV-cache append happen in this launch's epilogue)
  2. attn    [RoPE(q, k), KV-cache append unless done by 1.] softmax(q K^T / sqrt(d)) V     -> y
  3. wo      mask(tau_o) -> GEMV                                                            -> fp32 slabs
  4. gate|up [h = resid + round(sum wo-slabs); x = RMSNorm(h) * w] -> mask(tau_gate|up)      -> gate|up
  5. down    [x = silu(gate) * up] -> mask(tau_down) -> GEMV                                -> fp32 slabs
  lm_head    [h = resid + round(sum down-slabs); x = RMSNorm(h) * w] -> dense GEMV           -> logits

All through the C ABI (teal_fused_gemv / teal_decode_attention); PyTorch only owns the buffers and
the stream.  The rounding points are those of the reference's fp16/bf16 tensors, so logits agree
with the unfused module path to fp16 rounding (tests/test_engine.py).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional

import torch

from .. import _lib, runtime
from ..monkeypatch import UP_SHIFT_BYTES, to_column_major
from .model import Transformer

TEAL_IN_PLAIN, TEAL_IN_RESID_NORM, TEAL_IN_SILU_MUL, TEAL_IN_MASKED, TEAL_IN_ATTN_MERGE = 0, 1, 2, 3, 4
TEAL_OUT_ROUNDED, TEAL_OUT_SLABS, TEAL_OUT_PAIR_SILU, TEAL_OUT_QKV_ROPE, TEAL_OUT_SLAB_SUM = 0, 1, 2, 3, 4
MAX_SLABS = 32


class GemvIn(ctypes.Structure):  # teal_gemv_in_t
    _fields_ = [("mode", ctypes.c_int), ("x", ctypes.c_void_p), ("resid_in", ctypes.c_void_p),
                ("row_index", ctypes.c_void_p), ("slabs", ctypes.c_void_p), ("nslabs", ctypes.c_int),
                ("norm_weight", ctypes.c_void_p), ("eps", ctypes.c_float), ("resid_out", ctypes.c_void_p),
                ("masks", ctypes.c_void_p), ("att_head_dim", ctypes.c_int), ("att_nsplit", ctypes.c_int),
                ("slabs_interleaved", ctypes.c_int), ("gate_activated", ctypes.c_int)]


class GemvOut(ctypes.Structure):  # teal_gemv_out_t
    _fields_ = [("nseg", ctypes.c_int), ("w", ctypes.c_void_p * 3), ("ld", ctypes.c_int * 3),
                ("col0", ctypes.c_int * 3), ("ncols", ctypes.c_int * 3), ("tau", ctypes.c_float * 3),
                ("y", ctypes.c_void_p * 3), ("mode", ctypes.c_int), ("slabs", ctypes.c_void_p),
                ("slabs_bytes", ctypes.c_size_t), ("mask_out", ctypes.c_void_p), ("mask_tau", ctypes.c_float),
                ("slabs_interleaved", ctypes.c_int), ("weight_bits", ctypes.c_int), ("scale", ctypes.c_void_p * 3),
                ("scale_ld", ctypes.c_int * 3), ("groupsize", ctypes.c_int),
                ("rope", ctypes.c_void_p), ("rope_pos", ctypes.c_void_p), ("k_cache", ctypes.c_void_p), ("v_cache", ctypes.c_void_p),
                ("rope_head_dim", ctypes.c_int), ("rope_max_seq", ctypes.c_int), ("act_seg0", ctypes.c_int),
                ("desc", ctypes.c_char_p), ("desc_bytes", ctypes.c_int)]


def _out(segs, mode, slabs: Optional[torch.Tensor] = None) -> GemvOut:
    """segs: list of (weight_ptr, ld, col0, ncols, tau, y_ptr[, scale_ptr[, (scale_ld, groupsize)]]); a scale pointer
    marks quantised weights: int8 (per-column scales of the segment, element 0 = column col0) or, with the
    (scale_ld, groupsize) pair, int4 (the scales_and_zeros tensor of the whole image, col0 addressing both)."""
    o = GemvOut()
    o.nseg = len(segs)
    for i, seg in enumerate(segs):
        w, ld, col0, ncols, tau, y = seg[:6]
        o.w[i], o.ld[i], o.col0[i], o.ncols[i], o.tau[i], o.y[i] = w, ld, col0, ncols, tau, y
        if len(seg) > 6 and seg[6]:
            o.scale[i] = seg[6]
            o.weight_bits = 8
            if len(seg) > 7 and seg[7]:
                o.scale_ld[i], o.groupsize = seg[7]
                o.weight_bits = 4
    o.mode = mode
    if slabs is not None:
        o.slabs = slabs.data_ptr()
        o.slabs_bytes = slabs.numel() * 4
        o.slabs_interleaved = 1
    return o


class DecodeEngine:
    """One-token decode of `model` at batch 1 through the fused HIP kernels.

    `thresholds[i]` = {"q","k","v","o","gate","up","down": tau} for layer i (what monkeypatch_layer
    derives from the histograms, or the synthetic calibration).  The model's KV caches
    (`setup_caches`) are shared: prefill runs through the module path, decode through the engine.
    """

    @staticmethod
    def supports(model: Transformer, need_caches: bool = True) -> Optional[str]:
        """None if the fused step can run `model` as it stands, else the reason it cannot (the caller then keeps the
        op-by-op module path, which handles every shape torch does).  Mirrors the shape contracts of the launches:
        teal_decode_attention* (head_dim 64 / 128, [1][n_kv][max_seq][hd] caches), the register-resident RMSNorm
        producer (dim <= 16384), 16-byte rows (multiples of 8 columns), Z <= 65536, one weight format for all linears."""
        cfg = model.config
        lins = [lin for layer in model.layers for lin in (layer.attention.wqkv, layer.attention.wo, layer.feed_forward.w1,
                                                         layer.feed_forward.w3, layer.feed_forward.w2)] + [model.output]
        i4 = [hasattr(lin, "scales_and_zeros") for lin in lins]
        if any(i4):
            # int4 group-quantised projections (teal_amd/quantize.py) with a 16-bit lm_head, as quantize_model_int4 leaves them
            from ..quantize import int4_kernel_supports
            if not all(i4[:-1]) or i4[-1] or model.output.weight.dtype not in (torch.float16, torch.bfloat16):
                return "int4 blocks need every projection int4 group-quantised and the lm_head in fp16 / bf16"
            kvw = cfg.n_local_heads * cfg.head_dim
            if int(getattr(model, "tp_world", 1)) > 1:
                return "int4 group-quantised blocks are not sharded (tp.shard_linear)"
            for j, lin in enumerate(lins[:-1]):
                if not int4_kernel_supports(lin.in_features, lin.out_features, lin.groupsize, kvw if j % 5 == 0 else 0):
                    return f"int4 linear {lin.in_features}x{lin.out_features} g{lin.groupsize} is outside the int4 kernel's shape contract"
            i8, dt = [False] * len(lins), model.output.weight.dtype
        else:
            i8 = [lin.weight.dtype == torch.int8 for lin in lins]
            if any(i8) and not all(i8):
                return "mixed int8 / 16-bit linears"
            dt = model.output.scales.dtype if all(i8) else model.output.weight.dtype
            if dt not in (torch.float16, torch.bfloat16) or any((not b) and lin.weight.dtype != dt for b, lin in zip(i8, lins)):
                return f"weights are not uniformly fp16 / bf16 (or int8 with such scales): {dt}"
        if cfg.head_dim not in (64, 128):
            return f"head_dim {cfg.head_dim} (the attention launches are built for 64 and 128)"
        kv = cfg.n_local_heads * cfg.head_dim
        # under tensor parallelism (tp.apply_tp) cfg.n_head / n_local_heads and the projections' widths are the RANK's: q is
        # n_head * head_dim = dim / world wide, the residual stream stays dim wide
        world = int(getattr(model, "tp_world", 1))
        if cfg.dim % 64 or cfg.dim > 16384 or cfg.dim != cfg.n_head * cfg.head_dim * world:
            return f"dim {cfg.dim} (need a multiple of 64, <= 16384, = n_head * head_dim [* TP world])"
        inter = model.layers[0].feed_forward.w1.out_features
        if (cfg.n_head * cfg.head_dim) % 64:
            return f"rank-local query width {cfg.n_head * cfg.head_dim} must be a multiple of 64"
        if inter % 8 or inter > 65536 or kv % 8 or cfg.vocab_size % 8:
            return "intermediate_size / kv width / vocab_size must be multiples of 8 (intermediate_size <= 65536)"
        if not need_caches:  # a static verdict (shapes and weight formats), before setup_caches has run
            return None if model.output.weight.is_cuda else "model is not on a HIP device"
        if model.freqs_cis is None or model.freqs_cis.dtype != dt:
            return "caches are not set up (model.setup_caches) in the model dtype"
        for layer in model.layers:
            kc = getattr(layer.attention, "kv_cache", None)
            if kc is None or kc.k_cache.shape[0] != 1 or not kc.k_cache.is_contiguous() or not kc.v_cache.is_contiguous():
                return "KV caches must be contiguous with max_batch_size == 1"
        if not model.output.weight.is_cuda:
            return "model is not on a HIP device"
        return None

    def __init__(self, model: Transformer, thresholds: List[Dict[str, float]], pair: Optional[bool] = None,
                 att_split: int = 0, reduce_presummed: Optional[bool] = None):
        why = DecodeEngine.supports(model)
        if why is not None:
            raise ValueError(f"DecodeEngine cannot run this model: {why}")
        self.L = _lib.load()
        runtime.init()
        cfg = model.config
        self.cfg, self.model = cfg, model
        dev = model.output.weight.device
        dt = model.output.scales.dtype if hasattr(model.output, "scales") else model.output.weight.dtype
        lins = [lin for layer in model.layers for lin in (layer.attention.wqkv, layer.attention.wo, layer.feed_forward.w1,
                                                         layer.feed_forward.w3, layer.feed_forward.w2)] + [model.output]
        self.int4 = hasattr(lins[0], "scales_and_zeros")  # int4 group-quantised projections, 16-bit lm_head
        self.int8 = lins[0].weight.dtype == torch.int8    # int8 weight-only linears (teal_amd/quantize.py)
        self.dtype, self.code = dt, runtime.dtype_code(dt)
        assert model.freqs_cis is not None, "call model.setup_caches() first"
        # widths: `dim` = the residual stream (replicated under tensor parallelism), `qdim` / `kv` / `inter` = this rank's query,
        # key-value and intermediate columns (= the model's own without TP: tp.apply_tp divided cfg.n_head / n_local_heads and
        # sliced the linears, gpt-fast/tp.py:110-140)
        dim, hd = cfg.dim, cfg.head_dim
        inter = model.layers[0].feed_forward.w1.out_features
        qdim, kv = cfg.n_head * hd, cfg.n_local_heads * hd
        self.dim, self.qdim, self.inter, self.kv, self.nqkv = dim, qdim, inter, kv, qdim + 2 * kv
        for layer in model.layers:
            at_, ff_ = layer.attention, layer.feed_forward
            shapes = ((at_.wqkv.out_features, at_.wqkv.in_features), (at_.wo.out_features, at_.wo.in_features),
                      (ff_.w1.out_features, ff_.w1.in_features), (ff_.w3.out_features, ff_.w3.in_features),
                      (ff_.w2.out_features, ff_.w2.in_features))
            assert shapes == ((self.nqkv, dim), (dim, qdim), (inter, dim), (inter, dim), (dim, inter)), shapes
        # the block's two partial outputs (wo, down) summed over the ranks: reduce(fp32 slab buffer) in place, None = one rank
        self.reduce = getattr(model, "tp_reduce", None)
        self.tp_world = int(getattr(model, "tp_world", 1))
        self.gather = getattr(model, "tp_gather", None)  # all-gather of a 1-D fp32 tensor over the ranks (synthetic calibration)
        # reduce_presummed (TEAL_TP_PRESUM=1): wo / down fold their row slices themselves (TEAL_OUT_SLAB_SUM: arrival tickets, the
        # last slice of a tile adds the partials in slice order) and hand over ONE fp32 [dim] vector, so the ranks all-reduce
        # 16-32 KB — the reference's element count (gpt-fast/tp.py:120-121) in fp32 — instead of the [dim][4..8] slab buffer
        # (64-256 KB); costs the launch its ticket round (+1.2 us measured, profiles/r06_tp_rank_local_launches.txt).  Same
        # arithmetic on one rank (bit-identical: tests/test_rccl_one_rank.py); across ranks the sum associates (slices, then
        # ranks) instead of (ranks, then slices).  Off by default: the slab hand-over is the faster launch.
        if reduce_presummed is None:
            reduce_presummed = os.environ.get("TEAL_TP_PRESUM", "0") == "1"
        self.presum = bool(reduce_presummed)
        for layer in model.layers:
            for lin in (layer.attention.wqkv, layer.attention.wo, layer.feed_forward.w1, layer.feed_forward.w3,
                        layer.feed_forward.w2):
                if not self.int4:  # (an int4 linear holds the packed image of W^T already)
                    to_column_major(lin, shift_bytes=UP_SHIFT_BYTES if lin is layer.feed_forward.w3 else 0)
        to_column_major(model.output)
        e = lambda *shape, dtype=dt: torch.zeros(*shape, device=dev, dtype=dtype)  # noqa: E731
        self.resid = [e(dim), e(dim)]
        self.qkv, self.y_attn, self.gu = e(self.nqkv), e(qdim), e(2 * inter)
        self.h_mlp = e(inter)                                                  # silu(gate) * up (PAIR epilogue)
        self.y_mask = e((qdim + 63) // 64, dtype=torch.int64)                  # keep masks of y_attn vs tau_o
        self.h_mask = e((inter + 63) // 64, dtype=torch.int64)                 # keep masks of h_mlp vs tau_down
        # gate|up as one PAIR launch needs whole 64-column chunks and Z small enough for a single list
        can_pair = inter % 64 == 0 and dim % 64 == 0 and (dim + 1) * 4 <= 44 * 1024
        if self.int4:
            pair = False  # the int4 kernel has no paired form
        if pair is None and self.int8:
            pair = False  # int8 rows are half as long: the PAIR launch's 172 workgroups are request-bound
        if pair is None:
            # Pair (one launch streams the same column tile of gate AND up, silu * up + keep masks in its epilogue) only where
            # the paired tiles are 128 columns wide, i.e. where inter / 128 tiles still cover 2/3 of the CUs (Llama-2-70B).
            # Narrower models pair 64-column tiles = two 128-byte row segments per kept row, and the stream is bound by
            # row-segment REQUESTS: unpaired — two threshold segments of 128-column tiles (one 256-byte segment per row) and
            # a silu * up producer in the down launch — measured -1.4 % per token on Llama-2-7B, -0.3 % on Llama-3-8B, +0.8 %
            # on Llama-2-70B (layer_bench LB_PAIRAB, profiles/r03_layer_experiments.txt)
            ncu = int(self.L.teal_init())
            pair = inter * 3 >= 128 * ncu * 2
            # ... and where the unpaired launch would be a ragged second round of workgroups (a 16-wave workgroup owns its CU):
            # Llama-30B, 2 x 17920 / 128 = 280 tiles on 256 CUs ran 56.8 us = 0.53 of the HBM peak; paired, 140 tiles in one
            # round (round 6, profiles/r06_ratio_vs_width.txt)
            t_unpaired = 2 * ((inter + 127) // 128)
            pair = pair or ncu < t_unpaired < ncu * 3 // 2
        self.pair = bool(pair) and can_pair
        self.gate_act = False  # (set by _build: unpaired 16-bit / int8 gate | up stores silu(gate) | up)
        self.s_wo, self.s_down = e(MAX_SLABS, dim, dtype=torch.float32), e(MAX_SLABS, dim, dtype=torch.float32)
        self.s_qkv = e(8, self.nqkv, dtype=torch.float32)  # wqkv split-K slabs, summed by the attention launch
        if self.presum:
            if self.int4:
                raise ValueError("reduce_presummed: the int4 kernel has no TEAL_OUT_SLAB_SUM output")
            self.sum_wo, self.sum_down = e(dim, dtype=torch.float32), e(dim, dtype=torch.float32)
        self.logits = e(1, 1, cfg.vocab_size)
        self.ws = runtime.new_workspace(max(dim, inter), max(self.nqkv, inter, cfg.vocab_size))  # own header: own tickets
        self.rope = model.freqs_cis.contiguous()
        assert self.rope.dtype == dt and self.rope.shape[1:] == (hd // 2, 2)
        self.max_seq = model.max_seq_length
        # attention (flash-decoding): 4 or 8 workgroups per head write split-KV partials that the wo launch
        # merges in its prologue (no extra launch); very long contexts / wide models: up to 16 splits + a
        # merge launch.  att_split > 0 overrides (benchmark A/B).
        rep = cfg.n_head // cfg.n_local_heads
        if att_split:
            self.att_split = int(att_split)
        elif (rep == 8 and self.max_seq >= 2048) or (rep == 4 and self.max_seq >= 4096):
            # grouped-query kernel (teal_attention.hip: decode_attention_gqa_kernel): one workgroup per (KV head, split)
            # reads each K/V row once for the whole group; ~one workgroup per CU, more splits if the scores of a share
            # would not fit the LDS budget (kGqaMaxLds); merged by the merge launch (or by wo at 4 / 8 splits)
            ns = min(64, max(8, 256 // cfg.n_local_heads))
            while ns < 64 and self._gqa_lds_bytes(rep, hd, self.max_seq, ns) > 128 * 1024:
                ns *= 2
            self.att_split = ns
        elif self.max_seq <= 1024:
            self.att_split = 4
        elif self.max_seq <= 4096 and qdim <= 8192:
            self.att_split = 8
        elif self.max_seq <= 2048:
            self.att_split = 4
        else:
            self.att_split = min(16, max(2, (256 + cfg.n_head - 1) // cfg.n_head, (self.max_seq + 2047) // 2048))
        self.att_fused_merge = (self.att_split == 4 and qdim <= 16384) or (self.att_split == 8 and qdim <= 8192)
        # RoPE + KV-cache append in the epilogue of the wqkv launch (TEAL_OUT_QKV_ROPE: -1.3 % per token on Llama-2-7B @ 50 %,
        # profiles/r04_layer_experiments.txt) wherever that launch runs without split-K (the library reports which way it
        # went) and the per-query-head attention kernel follows (the grouped-query kernel of long contexts rotates itself)
        gqa_kernel = (rep == 8 and self.max_seq >= 2048) or (rep == 4 and self.max_seq >= 4096)
        self.rope_epilogue = bool(self.att_split) and not (self.int8 or self.int4) and not gqa_kernel
        self.att_ws = e(cfg.n_head * max(1, self.att_split) * (hd + 2), dtype=torch.float32)
        self.eps = float(cfg.norm_eps)
        self.n_wo = ctypes.c_int(0)
        self.n_down = ctypes.c_int(0)
        self.n_qkv = ctypes.c_int(0)
        self.rng_state = torch.tensor([1234, 0], dtype=torch.int64, device=dev)  # {seed, draw counter}
        self._seed, self._calls = 1234, 0
        self.token = torch.zeros(1, dtype=torch.int32, device=dev)
        self.tok_buf = torch.zeros(1, 1, dtype=torch.int32, device=dev)   # loop-carried token
        self.pos_buf = torch.zeros(1, dtype=torch.int32, device=dev)      # loop-carried position
        self.history = torch.zeros(max(8, model.max_seq_length), dtype=torch.int32, device=dev)
        self._graph, self._graph_key, self._graphs = None, None, {}
        self._build(thresholds)

    # ---- static launch descriptors (pointers never change: hipGraph-capture friendly) -------------
    def _build(self, ths):
        m, dim, qdim, inter, kv = self.model, self.dim, self.qdim, self.inter, self.kv
        hd_ = self.cfg.head_dim
        A, B = self.resid
        self.stages = []
        for i, layer in enumerate(m.layers):
            at, ff, th = layer.attention, layer.feed_forward, ths[i]
            # the row-wise projections' hand-over: interleaved fp32 slabs [dim][4 or 8], or (reduce_presummed) ONE fp32 [dim] vector
            il = 0 if self.presum else 1
            h_wo, h_down = (self.sum_wo, self.sum_down) if self.presum else (self.s_wo, self.s_down)

            def rowwise_out(lin, tau, dst):
                if not self.presum:
                    return _out([seg(lin, 0, dim, tau, None)], TEAL_OUT_SLABS, dst)
                o = _out([seg(lin, 0, dim, tau, None)], TEAL_OUT_SLAB_SUM)
                o.slabs, o.slabs_bytes = dst.data_ptr(), dst.numel() * 4
                return o

            k1_in = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=(m.tok_embeddings.weight.data_ptr() if i == 0 else A.data_ptr()),
                           slabs=(None if i == 0 else h_down.data_ptr()), nslabs=0, slabs_interleaved=il,
                           norm_weight=layer.attention_norm.weight.data_ptr(), eps=self.eps, resid_out=B.data_ptr())
            es = 2  # bytes per activation / scale element

            def sc(lin, col0=0):
                # int8: per-column scales from column col0; int4: the image's scales_and_zeros and (its columns per group
                # row, group size) — col0 of the segment addresses into it; 16-bit weights: nothing
                if self.int4:
                    return lin.scales_and_zeros.data_ptr(), (lin.out_features, lin.groupsize)
                return (lin.scales.data_ptr() + es * col0 if self.int8 else None), None

            def seg(lin, col0, ncols, tau, y):  # one threshold segment of a linear's weight image
                ld = lin.weight.stride(0) if self.int4 else lin.weight.stride(1)  # int4: bytes per row of the packed image
                return (lin.weight.data_ptr(), ld, col0, ncols, tau, y) + sc(lin, col0)

            k1_segs = [seg(at.wqkv, 0, qdim, th["q"], self.qkv.data_ptr()),
                       seg(at.wqkv, qdim, kv, th["k"], self.qkv.data_ptr() + 2 * qdim),
                       seg(at.wqkv, qdim + kv, kv, th["v"], self.qkv.data_ptr() + 2 * (qdim + kv))]
            # split attention: the projection writes fp32 slabs that the attention launch sums itself, so a narrow
            # (GQA) wqkv is row-sliced over all CUs without a reduce launch in between
            k1_out = _out(k1_segs, TEAL_OUT_SLABS, self.s_qkv) if self.att_split else _out(k1_segs, TEAL_OUT_ROUNDED)
            if self.rope_epilogue:
                kc0, vc0 = at.kv_cache.k_cache, at.kv_cache.v_cache
                k1_out.mode = TEAL_OUT_QKV_ROPE  # y[0] = rotated q; falls back to the slabs when the launch needs split-K
                k1_out.rope, k1_out.rope_pos = self.rope.data_ptr(), None  # (position pointer: per call, see _layer)
                k1_out.k_cache, k1_out.v_cache = kc0.data_ptr(), vc0.data_ptr()
                k1_out.rope_head_dim, k1_out.rope_max_seq = hd_, self.max_seq
            if self.att_fused_merge:
                k3_in = GemvIn(mode=TEAL_IN_ATTN_MERGE, x=self.att_ws.data_ptr(), att_head_dim=hd_, att_nsplit=self.att_split)
            elif self.pair:
                k3_in = GemvIn(mode=TEAL_IN_MASKED, x=self.y_attn.data_ptr(), masks=self.y_mask.data_ptr())
            else:
                k3_in = GemvIn(mode=TEAL_IN_PLAIN, x=self.y_attn.data_ptr())
            k3_out = rowwise_out(at.wo, th["o"], h_wo)
            k4_in = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=B.data_ptr(), slabs=h_wo.data_ptr(), nslabs=0, slabs_interleaved=il,
                           norm_weight=layer.ffn_norm.weight.data_ptr(), eps=self.eps, resid_out=A.data_ptr())
            k4_out = _out([seg(ff.w1, 0, inter, th["gate"], self.gu.data_ptr()),
                           seg(ff.w3, 0, inter, th["up"], self.gu.data_ptr() + 2 * inter)], TEAL_OUT_ROUNDED)
            if self.pair:
                k4_out = _out([seg(ff.w1, 0, inter, th["gate"], self.h_mlp.data_ptr()), seg(ff.w3, 0, inter, th["up"], None)],
                              TEAL_OUT_PAIR_SILU)
                k4_out.mask_out = self.h_mask.data_ptr()
                k4_out.mask_tau = th["down"]
                k5_in = GemvIn(mode=TEAL_IN_MASKED, x=self.h_mlp.data_ptr(), masks=self.h_mask.data_ptr())
            else:
                # unpaired: the gate tiles apply silu in their epilogue (act_seg0: once per column, by the two waves that
                # reduce the tile) and down's producer only multiplies — the activation out of the prologue of each of
                # down's 256 workgroups: -0.6 % per token on Llama-2-7B @ 50 % (profiles/r04_layer_experiments.txt)
                self.gate_act = not self.int4 and getattr(self, "use_gate_act", True)  # (use_gate_act = False: tests / A/B)
                k4_out.act_seg0 = 1 if self.gate_act else 0
                k5_in = GemvIn(mode=TEAL_IN_SILU_MUL, x=self.gu.data_ptr(), gate_activated=1 if self.gate_act else 0)
            k5_out = rowwise_out(ff.w2, th["down"], h_down)
            kc, vc = at.kv_cache.k_cache, at.kv_cache.v_cache
            assert kc.is_contiguous() and kc.shape[0] == 1 and kc.shape[2] == self.max_seq
            self.stages.append((k1_in, k1_out, kc, vc, k3_in, k3_out, k4_in, k4_out, k5_in, k5_out, th["o"]))
        self.head_in = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=A.data_ptr(), slabs=(self.sum_down if self.presum else self.s_down).data_ptr(),
                              nslabs=0, slabs_interleaved=0 if self.presum else 1,
                              norm_weight=m.norm.weight.data_ptr(), eps=self.eps, resid_out=None)
        self.head_out = _out([(m.output.weight.data_ptr(), m.output.weight.stride(1), 0, self.cfg.vocab_size, float("-inf"),
                               self.logits.data_ptr(), m.output.scales.data_ptr() if self.int8 else None)], TEAL_OUT_ROUNDED)

    def describe_launch(self, gin: GemvIn, gout: GemvOut, Z: int) -> str:
        """Kernel template instantiation and grid of the launch (gin, gout) makes — reported by the call itself through
        teal_gemv_out_t.desc (per call; the library keeps no "last launch" state).  Launches once, on the current stream."""
        buf = ctypes.create_string_buffer(160)
        keep = (gout.desc, gout.desc_bytes)
        gout.desc, gout.desc_bytes = ctypes.cast(buf, ctypes.c_char_p), 160
        try:
            self._stream = runtime.stream_ptr()
            self._gemv(gin, gout, Z, ctypes.c_int(0))
        finally:
            gout.desc, gout.desc_bytes = keep
        return buf.value.decode()

    def _gemv(self, gin: GemvIn, gout: GemvOut, Z: int, nslabs_out=None):
        rc = self.L.teal_fused_gemv(ctypes.byref(gin), ctypes.byref(gout), Z, self.code, self.ws.data_ptr(),
                                    self.ws.numel() * 4, ctypes.byref(nslabs_out) if nslabs_out is not None else None,
                                    self._stream)
        if rc != 0:
            _lib.check(rc, "teal_fused_gemv")

    def __call__(self, idx: torch.Tensor, input_pos: torch.Tensor, hook=None) -> torch.Tensor:
        """idx: int32 [1, 1] token id, input_pos: int32 [1] position -> logits [1, 1, vocab].

        `hook(when, stage, layer)` (tests / measurements only; never under graph capture) is called around every launch:
        when in {"before", "after"}, stage in {"qkv", "attn", "wo", "gate_up", "down", "head"}."""
        assert idx.dtype == torch.int32 and input_pos.dtype == torch.int32 and idx.numel() == 1
        self._stream = runtime.stream_ptr()
        tok_ptr, pos_ptr = idx.data_ptr(), input_pos.data_ptr()
        for i in range(len(self.stages)):
            self._layer(i, tok_ptr, pos_ptr, hook)
        self.head_in.nslabs = self.n_down.value
        if hook:
            hook("before", "head", -1)
        self._gemv(self.head_in, self.head_out, self.dim)
        if hook:
            hook("after", "head", -1)
        return self.logits

    @staticmethod
    def _gqa_lds_bytes(rep: int, hd: int, max_seq: int, nsplit: int) -> int:
        """LDS of the grouped-query attention launch (mirrors attention_split_impl in teal_attention.hip)."""
        nw = 8
        step = nw * (64 // (hd // 8))
        local = (((max_seq + step - 1) // step + nsplit - 1) // nsplit) * step
        return ((rep + 2) * (hd // 2) + 2 * rep * nw + rep * max(local, nw * hd)) * 4

    def _layer(self, i: int, tok_ptr: int, pos_ptr: int, hook=None, only=None):
        """The five launches of layer i (module docstring).  `only` (measurements: stage_times): launch just these stages, on
        whatever the buffers hold from the last full step."""
        cfg = self.cfg
        k1_in, k1_out, kc, vc, k3_in, k3_out, k4_in, k4_out, k5_in, k5_out, tau_o = self.stages[i]
        cb = hook if hook else (lambda *a: None)
        if only is not None:
            return self._layer_only(i, tok_ptr, pos_ptr, only)
        if i == 0:
            k1_in.row_index = tok_ptr
        else:
            k1_in.nslabs = self.n_down.value
        if self.rope_epilogue:
            k1_out.rope_pos = pos_ptr
        cb("before", "qkv", i)
        self._gemv(k1_in, k1_out, self.dim, self.n_qkv if self.att_split else None)
        cb("after", "qkv", i)
        ymask = self.y_mask.data_ptr() if self.pair else None
        cb("before", "attn", i)
        if self.att_split and self.rope_epilogue and self.n_qkv.value == 0:
            # the projection's epilogue rotated q (self.qkv[:dim]) and appended the token's k / v rows
            rc = self.L.teal_decode_attention_split_roped(self.qkv.data_ptr(), pos_ptr, kc.data_ptr(), vc.data_ptr(),
                                                          None if self.att_fused_merge else self.y_attn.data_ptr(), ymask, tau_o,
                                                          cfg.n_head, cfg.n_local_heads, cfg.head_dim, self.max_seq,
                                                          self.att_split, self.att_ws.data_ptr(), self.att_ws.numel() * 4,
                                                          self.code, self.ws.data_ptr(), self.ws.numel() * 4, self._stream)
        elif self.att_split:
            # (with y requested — long contexts / grouped-query shapes — a merge launch follows the split launch)
            rc = self.L.teal_decode_attention_split_ws(None, self.s_qkv.data_ptr(), self.n_qkv.value, self.rope.data_ptr(), pos_ptr,
                                                       kc.data_ptr(), vc.data_ptr(),
                                                       None if self.att_fused_merge else self.y_attn.data_ptr(), ymask, tau_o,
                                                       cfg.n_head, cfg.n_local_heads, cfg.head_dim, self.max_seq,
                                                       self.att_split, self.att_ws.data_ptr(), self.att_ws.numel() * 4,
                                                       self.code, self.ws.data_ptr(), self.ws.numel() * 4, self._stream)
        else:
            rc = self.L.teal_decode_attention_masked(self.qkv.data_ptr(), self.rope.data_ptr(), pos_ptr, kc.data_ptr(), vc.data_ptr(),
                                                     self.y_attn.data_ptr(), ymask, tau_o, cfg.n_head, cfg.n_local_heads,
                                                     cfg.head_dim, self.max_seq, self.code, self._stream)
        if rc != 0:
            _lib.check(rc, "teal_decode_attention")
        cb("after", "attn", i)
        cb("before", "wo", i)
        self._gemv(k3_in, k3_out, self.qdim, self.n_wo)
        if self.reduce is not None:
            self._reduce_slabs("wo")
        cb("after", "wo", i)
        k4_in.nslabs = self.n_wo.value
        cb("before", "gate_up", i)
        self._gemv(k4_in, k4_out, self.dim)
        cb("after", "gate_up", i)
        cb("before", "down", i)
        self._gemv(k5_in, k5_out, self.inter, self.n_down)
        if self.reduce is not None:
            self._reduce_slabs("down")
        cb("after", "down", i)

    def _layer_only(self, i: int, tok_ptr: int, pos_ptr: int, only):
        cfg = self.cfg
        k1_in, k1_out, kc, vc, k3_in, k3_out, k4_in, k4_out, k5_in, k5_out, tau_o = self.stages[i]
        if "qkv" in only:
            if i == 0:
                k1_in.row_index = tok_ptr
            else:
                k1_in.nslabs = self.n_down.value
            if self.rope_epilogue:
                k1_out.rope_pos = pos_ptr
            self._gemv(k1_in, k1_out, self.dim, self.n_qkv if self.att_split else None)
        if "attn" in only:
            assert self.att_split and self.att_fused_merge, "stage timing covers the split attention merged by wo"
            if self.rope_epilogue and self.n_qkv.value == 0:
                rc = self.L.teal_decode_attention_split_roped(self.qkv.data_ptr(), pos_ptr, kc.data_ptr(), vc.data_ptr(), None, None, tau_o,
                                                              cfg.n_head, cfg.n_local_heads, cfg.head_dim, self.max_seq, self.att_split,
                                                              self.att_ws.data_ptr(), self.att_ws.numel() * 4, self.code, self.ws.data_ptr(),
                                                              self.ws.numel() * 4, self._stream)
            else:
                rc = self.L.teal_decode_attention_split_ws(None, self.s_qkv.data_ptr(), self.n_qkv.value, self.rope.data_ptr(), pos_ptr,
                                                           kc.data_ptr(), vc.data_ptr(), None, None, tau_o, cfg.n_head, cfg.n_local_heads,
                                                           cfg.head_dim, self.max_seq, self.att_split, self.att_ws.data_ptr(),
                                                           self.att_ws.numel() * 4, self.code, self.ws.data_ptr(), self.ws.numel() * 4, self._stream)
            if rc != 0:
                _lib.check(rc, "teal_decode_attention")
        if "wo" in only:
            self._gemv(k3_in, k3_out, self.qdim, self.n_wo)
        if "gate_up" in only:
            k4_in.nslabs = self.n_wo.value
            self._gemv(k4_in, k4_out, self.dim)
        if "down" in only:
            self._gemv(k5_in, k5_out, self.inter, self.n_down)

    @torch.no_grad()
    def stage_times(self, reps: int = 20) -> Dict[str, float]:
        """Microseconds per launch, by stage: a hipGraph of that ONE stage over every layer (each layer's own weights: far
        beyond the Infinity Cache), HIP events around the replays, incl. the same-stream launch boundary; "layer" = the five
        launches of every layer in their real order.  Run after at least one full step (the buffers hold valid hand-overs).
        Measurement only: the KV row of the current position is rewritten with the same values."""
        assert self.reduce is None
        names = ("qkv", "attn", "wo", "gate_up", "down")
        out = {}
        tok_ptr, pos_ptr = self.tok_buf.data_ptr(), self.pos_buf.data_ptr()
        pos0 = self.pos_buf.clone()
        # algorithmic bytes per launch ON THE STATE THE STAGE'S GRAPH RUNS ON (every layer's launch consumes what the hand-over
        # buffers hold — a stage's replays do not modify its own inputs — with its own weights and thresholds): kept rows x row
        # bytes + the vectors in and out
        ths, wbytes = self.thresholds(), (1 if self.int8 else 2)
        dim, qd, kv, inter, L_ = self.dim, self.qdim, self.kv, self.inter, len(self.stages)

        def stage_bytes(key):
            ns_d, ns_w, ns_q = self.n_down.value, self.n_wo.value, self.n_qkv.value
            if key == "attn":
                return 2 * self.cfg.n_local_heads * (int(pos0.item()) + 1) * self.cfg.head_dim * 2
            tot = 0
            for i in range(L_):
                x = self._site_from_buffers(key, i, int(self.tok_buf.view(-1)[0]))
                if key == "qkv":
                    tot += sum(int((x > ths[i][p]).sum()) * n for p, n in (("q", qd), ("k", kv), ("v", kv))) * wbytes
                elif key == "gate_up":
                    tot += (int((x > ths[i]["gate"]).sum()) + int((x > ths[i]["up"]).sum())) * inter * wbytes
                else:
                    tot += int((x > ths[i]["o" if key == "wo" else "down"]).sum()) * dim * wbytes
            vec = {"qkv": dim * 2 + ns_d * dim * 4 + dim * 2 + (max(ns_q, 1) * self.nqkv * (4 if ns_q else 2)),
                   "wo": self.cfg.n_head * self.att_split * (self.cfg.head_dim + 2) * 4 + ns_w * dim * 4,
                   "gate_up": dim * 2 + ns_w * dim * 4 + dim * 2 + 2 * inter * 2,
                   "down": 2 * inter * 2 + ns_d * dim * 4}[key]
            return tot / L_ + vec

        out["bytes"] = {}
        for key in names + ("layer",):
            only = set(names) if key == "layer" else {key}

            def run():
                self._stream = runtime.stream_ptr()
                for i in range(len(self.stages)):
                    self._layer(i, tok_ptr, pos_ptr, only=only)

            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                run()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            if key != "layer":
                out["bytes"][key] = stage_bytes(key)
            g = torch.cuda.CUDAGraph()
            with runtime.graph_capture(g):
                run()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / len(self.stages))
            ts.sort()
            out[key] = ts[len(ts) // 2]
            del g
        self.pos_buf.copy_(pos0)
        return out

    def _reduce_slabs(self, which: str):
        """Tensor parallelism: the ONE sum over the ranks per attention ("wo") and per MLP ("down") (gpt-fast/tp.py:120-121,
        139-140), taken on the fp32 hand-over of the row-wise projection before anything is rounded: the consumer's RESID_NORM
        producer then computes h = resid + round(sum over slices AND ranks) with ONE rounding, exactly the unsharded step's
        expression (the reference all-reduces each rank's fp16-rounded output).  The hand-over is the interleaved slab buffer
        [dim][(n + 3) & ~3] — every rank launches the same geometry (a pure function of the local shape), so the buffers line up
        slab for slab — or, with reduce_presummed, the launch's own slice-order sum: one fp32 [dim]."""
        if self.presum:
            self.reduce(self.sum_wo if which == "wo" else self.sum_down)
            return
        slabs, n = (self.s_wo, self.n_wo.value) if which == "wo" else (self.s_down, self.n_down.value)
        self.reduce(slabs.view(-1)[: self.dim * ((n + 3) & ~3)])

    def reduce_bytes(self) -> Dict[str, int]:
        """payload of one all-reduce, by projection (after at least one step: the slab counts are the launches' own)"""
        if self.presum:
            return {"wo": self.dim * 4, "down": self.dim * 4}
        return {"wo": self.dim * ((self.n_wo.value + 3) & ~3) * 4, "down": self.dim * ((self.n_down.value + 3) & ~3) * 4}

    @torch.no_grad()
    def handover_sum(self, which: str) -> torch.Tensor:
        """fp32 [dim]: what the consumer's RESID_NORM producer adds up from the hand-over of `wo` / `down` as it stands — the
        slabs in slice order, or the presummed vector (measurements / restatements only)"""
        if self.presum:
            return (self.sum_wo if which == "wo" else self.sum_down).clone()
        slabs, n = (self.s_wo, self.n_wo.value) if which == "wo" else (self.s_down, self.n_down.value)
        st = (n + 3) & ~3
        v = slabs.view(-1)[: self.dim * st].view(self.dim, st)
        t = torch.zeros(self.dim, device=v.device, dtype=torch.float32)
        for j in range(n):
            t = t + v[:, j]
        return t

    @torch.no_grad()
    def site_activations(self, idx: torch.Tensor, input_pos: torch.Tensor) -> List[Dict[str, torch.Tensor]]:
        """|activation| every projection of every layer consumes in ONE decode step, restated with torch ops from the
        engine's own buffers while the step runs (sites as in the reference: q/k/v <- attention input, o <- attention
        output, gate/up <- MLP input, down <- silu(gate) * up).  Measurement / calibration only."""
        out: List[Dict[str, torch.Tensor]] = [dict() for _ in self.stages]

        def hook(when, stage, i):
            if i < 0 or when != "before":
                return
            site = {"qkv": "attn_in", "wo": "attn_out", "gate_up": "mlp_in", "down": "mlp_mid"}.get(stage)
            if site is not None:
                out[i][site] = self._site_from_buffers(stage, i, int(idx.view(-1)[0]))

        self(idx, input_pos, hook=hook)
        return out

    @torch.no_grad()
    def _site_from_buffers(self, stage: str, i: int, token: int = 0) -> torch.Tensor:
        """|activation| the launch `stage` of layer i would consume from what the hand-over buffers hold RIGHT NOW, restated
        with torch ops (residual + slabs -> RMSNorm; the split-KV merge; silu(gate) * up)."""
        dt, dim, inter = self.dtype, self.dim, self.inter
        A, B = self.resid
        layer = self.model.layers[i]

        def normed(resid, y, w):
            h = (resid + y).to(dt) if y is not None else resid
            hf = h.float()
            return ((hf * torch.rsqrt(hf.pow(2).mean() + self.eps)).to(dt) * w).float().abs()

        if stage == "qkv":
            if i == 0:
                resid, y = self.model.tok_embeddings.weight[token], None
            else:
                resid, y = A, self.handover_sum("down").to(dt)
            return normed(resid, y, layer.attention_norm.weight)
        if stage == "wo":
            if self.att_fused_merge:
                ns, hd = self.att_split, self.cfg.head_dim
                p = self.att_ws[: self.cfg.n_head * ns * (hd + 2)].view(self.cfg.n_head, ns, hd + 2)
                m, l, o = p[:, :, 0], p[:, :, 1], p[:, :, 2:]
                f = torch.where(l > 0, torch.exp(m - m.max(dim=1, keepdim=True).values), torch.zeros_like(m))
                return ((o * f[:, :, None]).sum(1) / (l * f).sum(1, keepdim=True)).reshape(-1).to(dt).float().abs()
            return self.y_attn.float().abs().clone()
        if stage == "gate_up":
            return normed(B, self.handover_sum("wo").to(dt), layer.ffn_norm.weight)
        assert stage == "down"
        if self.pair:
            return self.h_mlp.float().abs().clone()
        g, u = self.gu[:inter].float(), self.gu[inter:]
        sg = self.gu[:inter] if self.gate_act else torch.nn.functional.silu(g).to(dt)  # (act_seg0: silu already applied)
        return (sg * u).float().abs()

    SITE = {"q": "attn_in", "k": "attn_in", "v": "attn_in", "o": "attn_out", "gate": "mlp_in", "up": "mlp_in", "down": "mlp_mid"}

    def thresholds(self) -> List[Dict[str, float]]:
        """the thresholds the launch descriptors currently carry"""
        out = []
        for (_, k1_out, _, _, _, k3_out, _, k4_out, _, k5_out, _) in self.stages:
            out.append({"q": k1_out.tau[0], "k": k1_out.tau[1], "v": k1_out.tau[2], "o": k3_out.tau[0], "gate": k4_out.tau[0],
                        "up": k4_out.tau[1], "down": k5_out.tau[0]})
        return out

    @torch.no_grad()
    def kept_fractions(self, idx: torch.Tensor, input_pos: torch.Tensor) -> Dict[str, float]:
        """Achieved kept fraction of every projection's input on the decode activations of one step (mean over layers)."""
        acts, ths = self.site_activations(idx, input_pos), self.thresholds()
        return {p: sum(float((a[self.SITE[p]] > th[p]).float().mean()) for a, th in zip(acts, ths)) / len(acts) for p in self.SITE}

    @torch.no_grad()
    def _walk(self, first_token: torch.Tensor, pos0: int, n_steps: int, n_samples: int, visit):
        """n_steps eager decode steps from (first_token, pos0); at n_samples evenly spaced steps (first and last
        included) the step runs through site_activations and visit(acts) is called."""
        self.tok_buf.copy_(first_token.view(1, 1))
        self.pos_buf.fill_(pos0)
        n_samples = max(1, min(n_samples, n_steps))
        marks = {round(k * (n_steps - 1) / max(1, n_samples - 1)) for k in range(n_samples)}
        for t in range(n_steps):
            if t in marks:
                visit(self.site_activations(self.tok_buf, self.pos_buf))
                self.sample_fused(self.logits, 0.8, 200, feed=True)
            else:
                self._self_step(0.8, 200)

    @torch.no_grad()
    def calibrate_on_decode(self, sparsities: Dict[str, List[float]], first_token: torch.Tensor, pos0: int, n_steps: int,
                            n_samples: int = 5, rounds: int = 2) -> List[Dict[str, float]]:
        """Synthetic calibration on the DECODE path: thresholds = the target quantile of |activation| pooled over
        n_samples decode positions in [pos0, pos0 + n_steps), measured on the sparse path itself (`rounds` passes, since
        a projection's threshold shifts the activations downstream of it).  With random weights the attention output
        shrinks with the context length, so thresholds taken from a short prefill (generate.calibrate_thresholds) keep far
        fewer than the target of the o-projection's rows during a long decode; real checkpoints use the histograms."""
        assert pos0 + n_steps <= self.max_seq
        ths = self.thresholds()
        for _ in range(rounds):
            pool: List[Dict[str, List[torch.Tensor]]] = [dict(attn_in=[], attn_out=[], mlp_in=[], mlp_mid=[]) for _ in self.stages]

            def visit(acts):
                for i, a in enumerate(acts):
                    for k, v in a.items():
                        pool[i][k].append(v)

            self._walk(first_token, pos0, n_steps, n_samples, visit)
            for i in range(len(self.stages)):
                cat = {site: torch.cat(v) for site, v in pool[i].items()}
                if self.gather is not None:
                    # tensor parallelism: the row-wise projections' sites (attention output, silu(gate) * up) are SLICED over the
                    # ranks; the threshold is a property of the whole site, so every rank takes the quantile of the all-gathered
                    # samples — the unsharded model's own quantile — and every round already runs with it (the replicated
                    # sites hold the same samples on every rank: same reduce result, same bits)
                    for site in ("attn_out", "mlp_mid"):
                        cat[site] = self.gather(cat[site])
                for p, site in self.SITE.items():
                    s = float(sparsities[p][i])
                    ths[i][p] = -1.0 if s <= 0 else float(torch.quantile(cat[site], s))
            self._build(ths)
            self._graph = None
        return ths

    @torch.no_grad()
    def mean_kept_fractions(self, first_token: torch.Tensor, pos0: int, n_steps: int, n_samples: int = 3) -> Dict[str, float]:
        """kept fraction per projection, mean over layers and over n_samples decode positions of [pos0, pos0 + n_steps)"""
        acc = {p: [] for p in self.SITE}
        ths = self.thresholds()

        def visit(acts):
            for p in self.SITE:
                acc[p].append(sum(float((a[self.SITE[p]] > th[p]).float().mean()) for a, th in zip(acts, ths)) / len(acts))

        self._walk(first_token, pos0, n_steps, n_samples, visit)
        return {p: sum(v) / len(v) for p, v in acc.items()}

    def sample_fused(self, logits: torch.Tensor, temperature: float = 1.0, top_k: Optional[int] = None,
                     feed: bool = False) -> torch.Tensor:
        """One launch: top-k filter + softmax + exponential-race multinomial (generate.py:49-66).
        feed=True also carries the loop state on the device: the token goes into the buffer the next
        step reads, the position is advanced and the token is appended to `history`."""
        tok_out = self.tok_buf if feed else self.token
        rc = self.L.teal_sample_topk_ws(logits.data_ptr(), self.cfg.vocab_size, self.code, int(top_k or 0), float(temperature),
                                        self.rng_state.data_ptr(), tok_out.data_ptr(),
                                        self.pos_buf.data_ptr() if feed else None,
                                        self.history.data_ptr() if feed else None, self.history.numel(),
                                        self.ws.data_ptr(), self.ws.numel() * 4, runtime.stream_ptr())
        if rc != 0:
            _lib.check(rc, "teal_sample_topk")
        return tok_out

    def manual_seed(self, seed: int):
        self._seed, self._calls = int(seed), 0
        self.rng_state.copy_(torch.tensor([seed, 0], dtype=torch.int64))

    # ---- device-resident decode loop: one hipGraph replay == one token, no host-side glue ---------
    def _self_step(self, temperature, top_k):
        logits = self(self.tok_buf, self.pos_buf)
        self.sample_fused(logits, temperature, top_k, feed=True)

    def capture_loop(self, temperature: float, top_k: Optional[int], tokens: int = 1):
        """hipGraph of `tokens` consecutive decode steps (token, position and RNG counter stay on the device between them, so
        a graph can span any number of tokens; each replay costs one host launch and one graph boundary on the GPU)."""
        if self.reduce is not None and not getattr(self.reduce, "capturable", False):
            raise RuntimeError("this tensor-parallel engine's all-reduce cannot be captured into a hipGraph (gloo / host-staged); "
                               "decode eagerly (use_graph=False) or run the ranks over RCCL")
        key = (float(temperature), int(top_k or 0), int(tokens))
        if self._graph is not None and self._graph_key == key:
            return self._graph
        if self._graph is None:
            self._graphs = {}
        if key in self._graphs:
            self._graph, self._graph_key = self._graphs[key], key
            return self._graph
        state = (self.tok_buf.clone(), self.pos_buf.clone(), self.rng_state.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # warm-up outside capture (KV rows it writes are rewritten by the real run)
            self._self_step(temperature, top_k)
        torch.cuda.current_stream().wait_stream(s)
        self.tok_buf.copy_(state[0]); self.pos_buf.copy_(state[1]); self.rng_state.copy_(state[2])
        g = torch.cuda.CUDAGraph()
        try:
            with runtime.graph_capture(g):
                for _ in range(int(tokens)):
                    self._self_step(temperature, top_k)
        finally:  # (also when the capture fails — decode_n then decodes a sharded model eagerly from the same state)
            self.tok_buf.copy_(state[0]); self.pos_buf.copy_(state[1]); self.rng_state.copy_(state[2])
        self._graphs[key] = g
        self._graph, self._graph_key = g, key
        return g

    @torch.no_grad()
    def begin_sequence(self):
        """a new sample: its own random stream (seed + running count of samples), draw counter 0"""
        self._calls += 1
        self.rng_state.copy_(torch.tensor([self._seed + self._calls, 0], dtype=torch.int64))

    def sample_first(self, logits_row: torch.Tensor, temperature: float = 0.8, top_k: Optional[int] = 200) -> torch.Tensor:
        """the token after the prompt, drawn by the fused sampler from the prompt pass's last-position logits (one launch where
        the torch sampler of generate.sample takes ~10): opens the sample's random stream; follow with decode_n(..., drawn=1)"""
        assert logits_row.is_contiguous() and logits_row.numel() == self.cfg.vocab_size and logits_row.dtype == self.dtype
        self.begin_sequence()
        return self.sample_fused(logits_row, temperature, top_k, feed=False)

    def decode_n(self, first_token: torch.Tensor, pos: int, n: int, temperature: float = 0.8,
                 top_k: Optional[int] = 200, use_graph: bool = True, drawn: int = 0) -> torch.Tensor:
        """n decode steps starting from `first_token` at position `pos`; returns the n sampled tokens.  drawn: tokens already
        drawn from this sample's random stream (sample_first: 1) — the loop continues it instead of opening a new one."""
        assert n + drawn <= self.history.numel() and pos + n <= self.max_seq
        self.tok_buf.copy_(first_token.view(1, 1))
        self.pos_buf.fill_(pos)
        if not drawn:
            self.begin_sequence()
        if use_graph and self.reduce is not None and not getattr(self.reduce, "capturable", False):
            use_graph = False  # host-staged all-reduce (gloo): the step cannot live in a hipGraph
        if use_graph and getattr(self, "tp_capture_error", None):
            use_graph = False  # an earlier capture of this sharded step failed: stay eager
        g = None
        if use_graph:
            try:
                g = self.capture_loop(temperature, top_k)
            except Exception as e:  # noqa: BLE001
                # Under tensor parallelism the step holds two all-reduces per layer (RCCL, captured as graph nodes).  If this stack
                # cannot capture them, degrade to eager decode and say why instead of aborting the run; an unsharded step holds
                # only this library's launches — a capture failure there is a bug and propagates.
                if self.reduce is None:
                    raise
                self.tp_capture_error = f"{type(e).__name__}: {e}"
                print(f"teal_amd: hipGraph capture of the tensor-parallel decode step failed ({self.tp_capture_error}); decoding eagerly")
                torch.cuda.synchronize()
        if g is not None:
            for _ in range(n):
                g.replay()
        else:
            for _ in range(n):
                self._self_step(temperature, top_k)
        return self.history[drawn:drawn + n].clone()  # (the sampler files a token under its draw counter)

    # nn.Module-ish surface so GraphedDecoder can drive either a Transformer or an engine
    @property
    def config(self):
        return self.cfg

    @property
    def device(self):
        return self.logits.device


def make_engine_stepper(model: Transformer, a, ths: Optional[List[Dict[str, float]]] = None):
    """bench.py helper: thresholds (synthetic calibration) + prefill through the module path + the
    device-resident decode loop (one hipGraph replay per token); returns (step_fn, info).  `ths`: start from these
    thresholds on an already patched model (they are re-taken on the timed decode positions below) instead of the
    synthetic calibration from scratch, whose forward hooks need the un-patched modules."""
    from . import generate as G
    dev = "cuda"
    if ths is None:
        ths = G.apply_sparsity(model, sparsity=a.sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
    npr = int(getattr(a, "prompt_tokens", 6))
    prompt = torch.randint(0, model.config.vocab_size, (npr,), device=dev, dtype=torch.int,
                           generator=torch.Generator(device=dev).manual_seed(7))
    total = npr + 2 * (a.warmup + a.steps) + 16
    model.max_seq_length = -1
    model.setup_caches(max_batch_size=1, max_seq_length=min(total, model.config.block_size))
    with torch.no_grad():
        import time
        # the prompt pass as generate() runs it under --compile (prefill.FusedPrefill: the hand-fused HIP pass from a hipGraph for
        # prompts of up to 8 tokens, the module path otherwise); it fills the shared KV caches
        from .prefill import FusedPrefill
        G.relayout_for_engine(model)
        pre = FusedPrefill(model, graph=True)
        for _ in range(2):
            logits = pre(prompt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()  # timed warm (the reference's tok/s definition includes the prefill)
        logits = pre(prompt)
        torch.cuda.synchronize()
        prefill_s = time.perf_counter() - t0
        logits = logits.clone()
        tok = G.sample(logits, temperature=0.8, top_k=200)[0]
        cls, why = pick_engine(model)
        if cls is None:
            raise ValueError(f"no fused engine for this model: {why}")
        eng = cls(model, ths, att_split=int(getattr(a, "att_split", 0)), pair=getattr(a, "pair", None))
        span = min(a.warmup + a.steps + 4, eng.max_seq - npr)
        if a.sparsity > 0 and getattr(a, "decode_calibration", True):
            # thresholds for the kept fraction ON THE MEASURED DECODE POSITIONS (apply_sparsity's synthetic mode already
            # refined them on a generic 200-token decode; this takes them on exactly the timed range)
            sp = {p: [a.sparsity] * len(model.layers) for p in eng.SITE}
            ths = eng.calibrate_on_decode(sp, tok, npr, span)
        eng.tok_buf.copy_(tok.view(1, 1))
        eng.pos_buf.fill_(npr)
        U = max(1, int(getattr(a, "graph_tokens", 1)))
        graph_u = eng.capture_loop(0.8, 200, U) if U > 1 else None
        graph = eng.capture_loop(0.8, 200)
    # capture_loop replays the step a few times: read back where the stream stands
    state = {"pos": int(eng.pos_buf.item())}
    max_seq = eng.max_seq

    def step():
        # a run longer than the model's context (block_size) wraps back to the end of the prompt instead of
        # stepping past the KV cache; every step still attends over the positions it is at
        if state["pos"] + 1 >= max_seq:
            eng.pos_buf.fill_(npr)
            state["pos"] = npr
        graph.replay()
        state["pos"] += 1

    def run(n):
        """exactly n decode steps: replays of the U-token graph, the remainder one token at a time"""
        while n > 0:
            if graph_u is not None and n >= U and state["pos"] + U < max_seq:
                graph_u.replay()
                state["pos"] += U
                n -= U
            else:
                step()
                n -= 1

    step.run = run
    return step, {"thresholds": ths, "engine": eng, "prefill_s": prefill_s, "prefill_path": pre.used, "first_token": tok, "pos0": npr,
                  "span": span, "graph_tokens": U}


def pick_engine(model: Transformer, need_caches: bool = True):
    """(DecodeEngine, None) if the fused decode step can run `model` as it stands (16-bit, int8 or int4 group-quantised
    projections), else (None, reason): the caller keeps the op-by-op module path."""
    why = DecodeEngine.supports(model, need_caches)
    return (DecodeEngine, None) if why is None else (None, why)
