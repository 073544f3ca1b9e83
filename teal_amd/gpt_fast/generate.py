#!/usr/bin/env python3
"""Decode harness with the reference's CLI surface (gpt-fast/generate.py:528-558):

    python -m teal_amd.gpt_fast.generate --checkpoint_path .../model.pth --hist_path models/Llama-2-7B/histograms \
        --sparsity 0.5 --compile
    python -m teal_amd.gpt_fast.generate --synthetic 7B --sparsity 0.5 --compile          (no checkpoint needed)

Same flags (`--hist_path`, `--sparsity`, `--checkpoint_path`, `--compile`, `--num_samples`,
`--max_new_tokens`, `--top_k`, `--temperature`, `--prompt`, `--profile`), same tokens/sec definition
(generated tokens / wall time of one generate() call INCLUDING prefill, one warm-up sample then
`num_samples` timed ones, generate.py:431-506), same seed (1234).  Differences, all MI355X-first:

  * `--compile` means "capture the decode step into a hipGraph" (torch.cuda.CUDAGraph), not
    torch.compile/Inductor/Triton: the sparse GEMVs are hand-written HIP behind torch.ops.teal.*,
    launch geometry is fixed by shape, so there is nothing to trace or autotune.
  * `--precision {fp16,bf16}` (reference hard-codes fp16, generate.py:386), `--greedy_lookup DIR`
    (wires utils.get_layer_greedy_sparsities, which the reference imports but never calls),
    `--synthetic NAME` (random weights of the named architecture + thresholds calibrated on the
    synthetic activations, because there is no network for checkpoints here), `--engine`
    (fused HIP decode step, teal_amd/gpt_fast/engine.py) and `--dense` (no monkeypatch: baseline).
"""
from __future__ import annotations

import argparse
import collections
import contextlib
import itertools
import json
import os
import sys
import time
from pathlib import Path
from typing import Dict, List, Optional

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from teal_amd import runtime  # noqa: E402
from teal_amd.gpt_fast.model import ModelArgs, Transformer  # noqa: E402
from teal_amd.monkeypatch import monkeypatch_layer  # noqa: E402
from teal_amd.utils import PROJS, get_layer_greedy_sparsities  # noqa: E402

default_device = "cuda" if torch.cuda.is_available() else "cpu"


# ------------------------------------------------------------------------------------------------
# sampling (gpt-fast/generate.py:49-66: top-k, softmax, exponential-trick multinomial, no host sync)
# ------------------------------------------------------------------------------------------------
def multinomial_sample_one_no_sync(probs_sort: torch.Tensor) -> torch.Tensor:
    q = torch.empty_like(probs_sort).exponential_(1)
    return torch.argmax(probs_sort / q, dim=-1, keepdim=True).to(dtype=torch.int)


def logits_to_probs(logits: torch.Tensor, temperature: float = 1.0, top_k: Optional[int] = None) -> torch.Tensor:
    logits = logits / max(temperature, 1e-5)
    if top_k is not None:
        v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
        logits = torch.where(logits < v.select(-1, -1).unsqueeze(-1), -float("Inf"), logits)
    return torch.nn.functional.softmax(logits, dim=-1)


def sample(logits: torch.Tensor, temperature: float = 1.0, top_k: Optional[int] = None):
    probs = logits_to_probs(logits[0, -1].float(), temperature, top_k)
    return multinomial_sample_one_no_sync(probs), probs


# ------------------------------------------------------------------------------------------------
# model construction
# ------------------------------------------------------------------------------------------------
def build_synthetic_model(name: str, device: str, dtype: torch.dtype, seed: int = 1234, std: float = 0.02,
                          n_layer: Optional[int] = None, shard=None) -> Transformer:
    """Random-init weights N(0, std^2) at the exact shapes of the named architecture.

    `shard(model)` (tensor parallelism: tp.apply_tp) runs on the META model, before any storage exists: a rank then
    allocates only its slices, and fills each from the full random tensor of that parameter — drawn in the unsharded order
    from the same seed, so the ranks' weights ARE the slices of the unsharded synthetic model (one fp32 temporary of the
    largest parameter at a time)."""
    cfg = ModelArgs.from_name(name)
    if n_layer is not None:
        cfg.n_layer = n_layer
    with torch.device("meta"):
        model = Transformer(cfg).to(dtype)
    full_shape = {pname: tuple(p.shape) for pname, p in model.named_parameters()}
    if shard is not None:
        shard(model)
    g = torch.Generator(device=device).manual_seed(seed)
    # storage is allocated ONCE, in the target dtype, and filled in place: Llama-2-70B peaks at its 137 GB of weights plus
    # one fp32 temporary (it used to materialise the fp32 meta model first: 279 GB reserved of the GPU's 288 GB)
    model = model.to_empty(device=device)
    owner = {id(m.weight): m for m in model.modules() if hasattr(m, "_tp")}
    with torch.no_grad():
        for pname, p in model.named_parameters():
            if pname.endswith("norm.weight"):
                p.data.fill_(1.0)
                continue
            w = torch.randn(full_shape[pname], device=device, dtype=torch.float32, generator=g) * std
            if id(p) in owner:  # this rank's rows (column-wise) or columns (row-wise) of the full tensor
                style, ranges = owner[id(p)]._tp
                w = torch.cat([w.narrow(0 if style == "colwise" else 1, lo, hi - lo) for lo, hi in ranges], dim=0 if style == "colwise" else 1)
            p.data.copy_(w)
            del w
    return model.eval()


def load_checkpoint_model(checkpoint_path: Path, device: str, dtype: torch.dtype, shard=None) -> Transformer:
    """`shard(model)` (tensor parallelism: tp.apply_tp) runs on the memory-mapped CPU tensors right after the state dict is
    assigned — before anything moves to the device — like the reference (gpt-fast/generate.py:249-256)."""
    with torch.device("meta"):
        model = Transformer.from_name(checkpoint_path.parent.name)
    if "int8" in str(checkpoint_path):  # gpt-fast/generate.py:239-243: an int8 weight-only checkpoint (quantize.py --mode int8)
        from teal_amd.quantize import convert_for_runtime_int8
        print("Using int8 weight-only quantization!")
        convert_for_runtime_int8(model, dtype)
    int4 = "int4" in str(checkpoint_path)
    if int4:  # gpt-fast/generate.py:236-242: model_int4.g32.pth -> groupsize 32 (a checkpoint written by teal_amd.quantize:
        # quantize_model_int4(model).state_dict(); the reference's own int4 files hold a CUDA-only packed layout)
        from teal_amd.quantize import convert_for_runtime_int4
        groupsize = int(checkpoint_path.name.split(".")[-2][1:])
        print("Using int4 weight-only quantization!")
        convert_for_runtime_int4(model, groupsize)
    ckpt = torch.load(str(checkpoint_path), mmap=True, weights_only=True)
    if "model" in ckpt and "stories" in str(checkpoint_path):
        ckpt = ckpt["model"]
    model.load_state_dict(ckpt, assign=True)
    if shard is not None:
        shard(model)
    if int4:  # packed uint8 weights and bf16 group parameters stay as they are; everything else takes the activation dtype
        model = model.to(device=device)
        for prm in model.parameters():
            prm.data = prm.data.to(dtype)
        return model.eval()
    if "int8" in str(checkpoint_path):  # keep the int8 buffers int8: only floating tensors take the activation dtype
        model = model.to(device=device)
        for prm in list(model.parameters()) + [b for b in model.buffers() if b.is_floating_point()]:
            prm.data = prm.data.to(dtype)
        return model.eval()
    return model.to(device=device, dtype=dtype).eval()


@torch.no_grad()
def calibrate_thresholds(model: Transformer, sparsities: Dict[str, List[float]], n_tokens: int = 24,
                         seed: int = 4321) -> List[Dict[str, float]]:
    """Synthetic mode: per layer and projection, tau = the `s` quantile of |activation| observed on a
    short dense run, so the kept fraction is 1 - s on the synthetic activations (the calibration
    histograms only describe real checkpoints).  Activation sites as in the reference:
    q/k/v <- attention input, o <- attention output, gate/up <- MLP input, down <- silu(g)*u."""
    if all(float(v) <= 0 for vals in sparsities.values() for v in vals):
        return [{p: -1.0 for p in PROJS} for _ in model.layers]
    dev = model.output.weight.device
    acts: List[Dict[str, List[torch.Tensor]]] = [dict(attn_in=[], attn_out=[], mlp_in=[], mlp_mid=[]) for _ in model.layers]
    hooks = []
    for i, layer in enumerate(model.layers):
        hooks.append(layer.attention.register_forward_pre_hook(lambda m, a, i=i: acts[i]["attn_in"].append(a[0].detach().float().flatten())))
        hooks.append(layer.attention.wo.register_forward_pre_hook(lambda m, a, i=i: acts[i]["attn_out"].append(a[0].detach().float().flatten())))
        hooks.append(layer.feed_forward.register_forward_pre_hook(lambda m, a, i=i: acts[i]["mlp_in"].append(a[0].detach().float().flatten())))
        hooks.append(layer.feed_forward.w2.register_forward_pre_hook(lambda m, a, i=i: acts[i]["mlp_mid"].append(a[0].detach().float().flatten())))
    g = torch.Generator(device=dev).manual_seed(seed)
    toks = torch.randint(0, model.config.vocab_size, (n_tokens,), device=dev, generator=g, dtype=torch.int)
    model.setup_caches(max_batch_size=1, max_seq_length=max(n_tokens, 8))
    model(toks.view(1, -1), torch.arange(n_tokens, device=dev))  # one dense prefill = n_tokens activation samples
    for h in hooks:
        h.remove()
    site = {"q": "attn_in", "k": "attn_in", "v": "attn_in", "o": "attn_out", "gate": "mlp_in", "up": "mlp_in", "down": "mlp_mid"}
    out = []
    for i in range(len(model.layers)):
        th = {}
        for p in PROJS:
            s = float(sparsities[p][i])
            if s <= 0:
                th[p] = -1.0  # keep everything (|x| > -1): the dense comparator on the same kernels
                continue
            a = torch.cat(acts[i][site[p]]).abs()
            th[p] = float(torch.quantile(a[:: max(1, a.numel() // 200000)], s))
        out.append(th)
    # the caches were sized for calibration; let generate() size them again
    model.max_seq_length = -1
    model.max_batch_size = -1
    return out


@torch.no_grad()
def refine_thresholds_on_decode(model: Transformer, sparsities: Dict[str, List[float]], ths: List[Dict[str, float]],
                                n_prompt: int = 24, n_decode: int = 200) -> List[Dict[str, float]]:
    """Synthetic mode, GPU: re-take the thresholds on the DECODE path (engine.calibrate_on_decode) so that every
    projection keeps its target fraction on decode activations — with random weights the attention output shrinks with
    the context length, and thresholds from a short prefill keep ~10 % of the o-projection's rows after 100 positions.
    The refined values are written back into the blocks' thresh_* attributes; the caches are released again."""
    dev = model.output.weight.device
    span = min(n_decode, model.config.block_size - n_prompt - 8)
    if span < 8:
        return ths
    model.max_seq_length, model.max_batch_size = -1, -1
    model.setup_caches(max_batch_size=1, max_seq_length=n_prompt + span + 8)
    from teal_amd.gpt_fast.engine import pick_engine
    cls, _why = pick_engine(model)
    if cls is None:  # e.g. head_dim 48: the module path decodes it, with the prefill thresholds
        model.max_seq_length, model.max_batch_size = -1, -1
        return ths
    toks = torch.randint(0, model.config.vocab_size, (n_prompt,), device=dev, dtype=torch.int,
                         generator=torch.Generator(device=dev).manual_seed(97))
    model(toks.view(1, -1), torch.arange(n_prompt, device=dev))
    eng = cls(model, ths)
    ths = eng.calibrate_on_decode(sparsities, toks[-1:].clone(), n_prompt, span)
    from teal_amd.gpt_fast import tp
    ths = tp.sync_thresholds(ths, model)
    for layer, th in zip(model.layers, ths):
        at, ff = layer.attention, layer.feed_forward
        at.thresh_q, at.thresh_k, at.thresh_v, at.thresh_o = th["q"], th["k"], th["v"], th["o"]
        ff.thresh_gate, ff.thresh_up, ff.thresh_down = th["gate"], th["up"], th["down"]
    del eng
    model.max_seq_length, model.max_batch_size = -1, -1  # let generate() size the caches again
    return ths


def apply_sparsity(model: Transformer, *, sparsity: float, hist_path: Optional[str], greedy_lookup: Optional[str],
                   synthetic: bool, decode_calibration: bool = True) -> List[Dict[str, float]]:
    """monkeypatch every layer (gpt-fast/generate.py:328-331); returns the thresholds used."""
    L = len(model.layers)
    if greedy_lookup and greedy_lookup.endswith(".json"):
        # block-wise greedy table captured from the reference (tests/golden/greedy_*.json, generated by
        # oracle/gen_golden.py from models/<name>/lookup): {"targets": {"0.5": {"sparsities": {proj: [...]}}}}
        with open(greedy_lookup) as f:
            table = json.load(f)["targets"][repr(float(sparsity))]["sparsities"]
        sparsities = {p: [float(v) for v in table[p][:L]] for p in PROJS}
        assert all(len(v) == L for v in sparsities.values()), "greedy table has fewer layers than the model"
    elif greedy_lookup:
        sparsities = get_layer_greedy_sparsities([sparsity] * L, greedy_lookup)
    else:
        sparsities = {p: [sparsity] * L for p in PROJS}
    device = model.output.weight.device.type
    if synthetic or hist_path is None:
        from teal_amd.gpt_fast import tp
        ths = tp.sync_thresholds(calibrate_thresholds(model, sparsities), model)  # (one tau per site on every rank; no-op without TP)
        for i, layer in enumerate(model.layers):
            monkeypatch_layer(i, layer, sparsity, None, device, thresholds=ths[i])
        if decode_calibration and device == "cuda" and any(float(v) > 0 for vals in sparsities.values() for v in vals):
            ths = refine_thresholds_on_decode(model, sparsities, ths)
    else:
        ths = [monkeypatch_layer(i, layer, sparsity, hist_path, device, sparsities=sparsities)
               for i, layer in enumerate(model.layers)]
    return ths


@torch.no_grad()
def report_kept_fractions(model: Transformer, thresholds, prompt: torch.Tensor) -> Dict[str, float]:
    """Achieved kept fraction per projection (the quantity that determines the bytes read), measured on
    one pass of the prompt through the patched model with the installed thresholds."""
    acts = {k: [] for k in ("attn_in", "attn_out", "mlp_in", "mlp_mid")}
    site = {"q": "attn_in", "k": "attn_in", "v": "attn_in", "o": "attn_out", "gate": "mlp_in", "up": "mlp_in", "down": "mlp_mid"}
    kept = {p: [] for p in PROJS}
    hooks = []
    for i, layer in enumerate(model.layers):
        th = thresholds[i]

        def pre_attn(m, a, th=th):
            x = a[0].detach().float().abs()
            for p in ("q", "k", "v"):
                kept[p].append(float((x > th[p]).float().mean()))

        def pre_ffn(m, a, th=th):
            x = a[0].detach().float().abs()
            for p in ("gate", "up"):
                kept[p].append(float((x > th[p]).float().mean()))

        hooks.append(layer.attention.register_forward_pre_hook(pre_attn))
        hooks.append(layer.feed_forward.register_forward_pre_hook(pre_ffn))
    model(prompt.view(1, -1), torch.arange(prompt.numel(), device=prompt.device))
    for h in hooks:
        h.remove()
    out = {p: sum(v) / len(v) for p, v in kept.items() if v}
    print("achieved kept fraction (prompt activations):", {k: round(v, 3) for k, v in out.items()})
    return out


def _get_model_size(model) -> int:
    size = 0
    for _, child in model.named_children():
        if not isinstance(child, torch.nn.Embedding):
            size += sum(p.numel() * p.dtype.itemsize for p in itertools.chain(child.parameters(), child.buffers()))
    return size


# ------------------------------------------------------------------------------------------------
# decode step + hipGraph capture
# ------------------------------------------------------------------------------------------------
class GraphedDecoder:
    """decode_one_token (gpt-fast/generate.py:73-77) with static buffers, captured once into a hipGraph
    (the reference gets the same effect from torch.compile(mode="reduce-overhead"), generate.py:420)."""

    def __init__(self, model: Transformer, use_graph: bool, temperature: float, top_k: Optional[int]):
        self.model, self.use_graph = model, use_graph
        self.kw = dict(temperature=temperature, top_k=top_k)
        dev = model.device if hasattr(model, "device") else model.output.weight.device
        self.tok = torch.zeros(1, 1, dtype=torch.int, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int, device=dev)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out_tok = None

    def _step(self):
        logits = self.model(self.tok, self.pos)
        if hasattr(self.model, "sample_fused"):  # HIP engine: one-launch sampler
            return self.model.sample_fused(logits, **self.kw)
        return sample(logits, **self.kw)[0]

    def capture(self, cur_token: Optional[torch.Tensor] = None, input_pos: Optional[torch.Tensor] = None):
        """cur_token / input_pos: where decoding stands.  The two eager warm-up steps and the capture run write the KV
        row of `input_pos` (rewritten with the same values by the first replay) — without them they would run at the
        stale buffers' position (0 after construction) and overwrite the first prompt token's K/V."""
        if not self.use_graph or self.graph is not None:
            return
        if cur_token is not None:
            self.tok.copy_(cur_token.view(1, 1))
        if input_pos is not None:
            self.pos.copy_(input_pos)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # warm-up outside capture: workspace + allocator pools
            for _ in range(2):
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with runtime.graph_capture(self.graph):
            self.out_tok = self._step()

    def __call__(self, cur_token: torch.Tensor, input_pos: torch.Tensor) -> torch.Tensor:
        self.tok.copy_(cur_token.view(1, 1))
        self.pos.copy_(input_pos)
        if self.graph is not None:
            self.graph.replay()
            return self.out_tok
        return self._step()


class GraphedPrefill:
    """`--compile_prefill` (gpt-fast/generate.py:423-425, 540): the prompt pass captured into a hipGraph per prompt
    length (the reference compiles `prefill` with Inductor).  Static token / position buffers, kept alive with the
    graph; the model's KV caches must not be re-allocated between capture and replay (setup_caches keeps them
    when the sizes are unchanged; a new cache object gets a new graph)."""

    MAX_GRAPHS = 4  # distinct (prompt length, buffers) keys kept; each graph owns a private memory pool (--interactive with
                    # varying prompt lengths would otherwise grow one per new length forever): least recently used goes first

    def __init__(self, model: Transformer):
        self.model = model
        self.graphs = collections.OrderedDict()
        self.eager_reason = None  # set when a capture failed under tensor parallelism: the pass then runs op by op

    def __call__(self, prompt: torch.Tensor) -> torch.Tensor:
        T = prompt.numel()
        if self.eager_reason is not None:
            return self.model(prompt.view(1, -1), torch.arange(0, T, device=prompt.device))
        # everything the captured launches hold raw pointers to: a re-laid-out weight (DecodeEngine / monkeypatch
        # to_column_major replace the storage) or a re-allocated KV cache forces a new capture
        key = (T, self.model.max_seq_length, self.model.output.weight.data_ptr(), self.model.tok_embeddings.weight.data_ptr()) + tuple(
            p for layer in self.model.layers for p in (layer.attention.kv_cache.k_cache.data_ptr(), layer.attention.kv_cache.v_cache.data_ptr(),
                                                       layer.attention.wqkv.weight.data_ptr(), layer.attention.wo.weight.data_ptr(),
                                                       layer.feed_forward.w1.weight.data_ptr(), layer.feed_forward.w2.weight.data_ptr(),
                                                       layer.feed_forward.w3.weight.data_ptr()))
        if key not in self.graphs:
            dev = prompt.device
            toks = torch.zeros(1, T, dtype=torch.int, device=dev)
            pos = torch.arange(0, T, device=dev)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                toks.copy_(prompt.view(1, -1))
                self.model(toks, pos)  # warm-up outside capture (allocator pools, GEMM heuristics)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            try:
                with runtime.graph_capture(g):
                    logits = self.model(toks, pos)
            except Exception as e:  # noqa: BLE001
                # a sharded model's pass holds one all-reduce per attention and per MLP (tp._reduce_hook): if the collective
                # cannot be captured on this stack, run the pass op by op instead of aborting the run (no silent change for an
                # unsharded model: there a capture failure is a bug and propagates)
                if int(getattr(self.model, "tp_world", 1)) < 2:
                    raise
                self.eager_reason = f"{type(e).__name__}: {e}"
                print(f"teal_amd: hipGraph capture of the tensor-parallel prompt pass failed ({self.eager_reason}); running it eagerly")
                torch.cuda.synchronize()
                return self.model(prompt.view(1, -1), torch.arange(0, T, device=dev))
            self.graphs[key] = (g, toks, pos, logits)  # every tensor the graph reads or writes stays alive
            while len(self.graphs) > self.MAX_GRAPHS:
                torch.cuda.synchronize()  # the evicted graph's pool goes back to the allocator: nothing of it may be in flight
                self.graphs.popitem(last=False)
        self.graphs.move_to_end(key)
        g, toks, _pos, logits = self.graphs[key]
        toks.copy_(prompt.view(1, -1))
        g.replay()
        return logits


class EngineDecoder:
    """GraphedDecoder's role for the fused HIP engine: built lazily once the KV caches exist."""

    def __init__(self, torch_model: Transformer, thresholds, use_graph: bool, temperature: float, top_k: Optional[int]):
        self.torch_model, self.thresholds, self.use_graph = torch_model, thresholds, use_graph
        self.kw = dict(temperature=temperature, top_k=top_k)
        self._engine, self._key = None, None

    def _cache_key(self):
        m = self.torch_model
        return (m.max_seq_length,) + tuple(p for layer in m.layers for p in (layer.attention.kv_cache.k_cache.data_ptr(),
                                                                              layer.attention.kv_cache.v_cache.data_ptr()))

    @property
    def model(self):
        # the engine's launch descriptors hold the KV caches' raw pointers: rebuild when setup_caches re-allocated them
        # (same size or not), not only when the context length changed
        key = self._cache_key()
        if self._engine is None or self._key != key:
            from teal_amd.gpt_fast.engine import pick_engine
            cls, why = pick_engine(self.torch_model)
            if cls is None:
                raise ValueError(f"no fused engine for this model: {why}")
            self._engine, self._key = cls(self.torch_model, self.thresholds), key
        return self._engine


def relayout_for_engine(model: Transformer) -> None:
    """The fused engine re-lays every projection (and lm_head) out column-major when it is built — lazily, after the first
    prefill.  Do it up front so that a prefill graph never captures pointers to storage that is freed later."""
    from teal_amd.monkeypatch import UP_SHIFT_BYTES, to_column_major
    for layer in model.layers:
        for lin in (layer.attention.wqkv, layer.attention.wo, layer.feed_forward.w1, layer.feed_forward.w3, layer.feed_forward.w2):
            if not hasattr(lin, "scales_and_zeros"):  # (an int4 image is packed column-gathered already)
                to_column_major(lin, shift_bytes=UP_SHIFT_BYTES if lin is layer.feed_forward.w3 else 0)
    to_column_major(model.output)


@torch.no_grad()
def generate(model: Transformer, prompt: torch.Tensor, max_new_tokens: int, decoder: GraphedDecoder,
             temperature: float = 0.8, top_k: Optional[int] = 200, prefill: Optional[GraphedPrefill] = None) -> torch.Tensor:
    """prefill (seq > 1: ops fall back to dense matmul) then max_new_tokens-1 decode steps."""
    T = prompt.size(0)
    T_new = T + max_new_tokens
    dev = prompt.device
    model.setup_caches(max_batch_size=1, max_seq_length=min(T_new, model.config.block_size))
    seq = torch.empty(T_new, dtype=prompt.dtype, device=dev)
    seq[:T] = prompt
    logits = prefill(prompt) if prefill is not None else model(prompt.view(1, -1), torch.arange(0, T, device=dev))
    eng = decoder.model if hasattr(decoder.model, "decode_n") else None
    if eng is not None and logits.dtype == eng.dtype and logits.shape[-1] == eng.cfg.vocab_size:
        # HIP engine: the token after the prompt comes from the same fused sampler as every later one (gpt-fast/generate.py:
        # 49-66 is ONE function for both), and the whole loop stays on the device
        next_token = eng.sample_first(logits[0, -1], decoder.kw["temperature"], decoder.kw["top_k"])
        seq[T] = next_token.view(())
        toks = eng.decode_n(next_token, T, max_new_tokens - 1, temperature=decoder.kw["temperature"],
                            top_k=decoder.kw["top_k"], use_graph=decoder.use_graph, drawn=1)
        seq[T + 1:] = toks.to(seq.dtype)
        return seq
    next_token = sample(logits, temperature=temperature, top_k=top_k)[0].clone()
    seq[T] = next_token
    if eng is not None:
        toks = eng.decode_n(next_token, T, max_new_tokens - 1, temperature=decoder.kw["temperature"],
                            top_k=decoder.kw["top_k"], use_graph=decoder.use_graph)
        seq[T + 1:] = toks.to(seq.dtype)
        return seq
    input_pos = torch.tensor([T], device=dev, dtype=torch.int)
    cur = next_token.view(1, -1)
    decoder.capture(cur, input_pos)
    for i in range(max_new_tokens - 1):
        nxt = decoder(cur, input_pos)
        input_pos += 1
        seq[T + 1 + i] = nxt.view(())
        cur = nxt.view(1, -1)
    return seq


# ------------------------------------------------------------------------------------------------
def main(args) -> Dict:
    device = args.device
    assert "cuda" in device, "the sparse decode path is GPU-only (HIP kernels, no CPU fallback)"
    if getattr(args, "draft_checkpoint_path", None) is not None:
        raise SystemExit("--draft_checkpoint_path: speculative decoding is not part of this build (the reference lists it as "
                         "untested with TEAL); run without a draft model")
    if getattr(args, "interactive", False) and args.synthetic:
        raise SystemExit("--interactive needs a tokenizer (a checkpoint directory), not --synthetic")
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.precision]
    # tensor parallelism as in the reference (gpt-fast/generate.py:249-256, tp.py): one process per GPU under
    # torch.distributed.run; FIRST, so that every rank builds / loads its shard on its own device.  A single process (the
    # north-star configuration) is untouched
    from teal_amd.gpt_fast import tp
    tp_rank = tp.maybe_init_dist()
    shard = None
    if tp_rank is not None:
        device = f"cuda:{torch.cuda.current_device()}"
        shard = tp.apply_tp  # wqkv / w1 / w3 column-wise, wo / w2 row-wise, one all-reduce per attention and per MLP
        if tp_rank != 0:
            import builtins
            builtins.print = lambda *a, **k: None  # rank 0 reports (tp.py's `print` override)
    from teal_amd import runtime
    runtime.init()
    t0 = time.time()
    if args.synthetic:
        model = build_synthetic_model(args.synthetic, device, dtype, n_layer=args.n_layer, shard=shard)
        prompt = torch.randint(0, model.config.vocab_size, (6,), device=device, dtype=torch.int,
                               generator=torch.Generator(device=device).manual_seed(7))  # "Hello, my name is" + BOS = 6 ids
        tokenizer = None
    else:
        assert args.checkpoint_path.is_file(), args.checkpoint_path
        model = load_checkpoint_model(args.checkpoint_path, device, dtype, shard=shard)
        from teal_amd.gpt_fast.tokenizer import get_tokenizer
        tokenizer = get_tokenizer(args.checkpoint_path.parent / "tokenizer.model", args.checkpoint_path)
        prompt = torch.tensor([tokenizer.bos_id()] + tokenizer.encode(args.prompt), dtype=torch.int, device=device)
    thresholds = None
    if not args.dense and (args.hist_path is not None or args.synthetic):
        # like the reference, patching is gated on hist_path, not on sparsity (generate.py:328)
        print("Monkeypatching with activation sparsity...")
        thresholds = apply_sparsity(model, sparsity=args.sparsity, hist_path=args.hist_path,
                                    greedy_lookup=args.greedy_lookup, synthetic=bool(args.synthetic))
    if getattr(args, "no_fused_decode", False):
        model.fused_decode = False
    torch.cuda.synchronize()
    print(f"Time to load model: {time.time() - t0:.02f} seconds")
    torch.manual_seed(1234)
    model_size = _get_model_size(model)
    # a hipGraph holds the step only if the ranks' all-reduce can be captured (RCCL); a host-staged gloo reduce decodes eagerly
    # (TEAL_TP_GRAPH=0 keeps a sharded model's decode eager even over RCCL; a capture that fails falls back by itself)
    use_graph = args.compile and (tp_rank is None or (bool(getattr(getattr(model, "tp_reduce", None), "capturable", False))
                                                      and os.environ.get("TEAL_TP_GRAPH", "1") != "0"))
    decoder = GraphedDecoder(model, use_graph, args.temperature, args.top_k)
    use_engine = args.engine or (args.compile and thresholds is not None and not getattr(args, "no_engine", False))
    if use_engine and not args.engine:
        # --compile implies the device-resident engine loop only for models the fused step can run (int4 blocks, head_dim 48,
        # ... decode through the patched modules under the same hipGraph capture instead)
        from teal_amd.gpt_fast.engine import pick_engine
        _, why = pick_engine(model, need_caches=False)
        if why is not None:
            print(f"fused engine not used: {why}")
            use_engine = False
    if use_engine:
        assert thresholds is not None, "--engine needs thresholds (--hist_path or --synthetic)"
        decoder = EngineDecoder(model, thresholds, use_graph, args.temperature, args.top_k)
        relayout_for_engine(model)
    # --compile captures the prompt pass as well (one hipGraph per prompt length; the reference compiles `prefill` only under
    # --compile_prefill, generate.py:423-425 — its Inductor compile takes minutes, a capture here takes milliseconds, and the
    # eager pass is ~400 launches = nine decode steps' worth for a 6-token prompt).  The prefill stays DENSE
    # (kernels/sparse_gemv.py:271,298).  --eager_prefill keeps the op-by-op pass; a host-staged TP all-reduce cannot be captured.
    want_graph_prefill = (getattr(args, "compile_prefill", False) or args.compile) and not getattr(args, "eager_prefill", False)
    prefill = GraphedPrefill(model) if (want_graph_prefill and use_graph) else None
    if thresholds is not None and want_graph_prefill and not getattr(args, "module_prefill", False):
        # prompts of up to 8 tokens: the hand-fused HIP prompt pass (teal_amd/gpt_fast/prefill.py: eight launches per layer over
        # the decode step's weight images, dense); longer prompts, quantised or sharded models: the pass above
        from teal_amd.gpt_fast.prefill import FusedPrefill
        prefill = FusedPrefill(model, graph=use_graph, fallback=prefill)
    tps, seqs = [], []
    start = -1 if args.compile else 0
    for i in range(start, args.num_samples):
        if getattr(args, "interactive", False) and i >= 0:
            text = input("What is your prompt? ")
            if "chat" in str(args.checkpoint_path):
                text = f"[INST] {text.strip()} [/INST]"
            prompt = torch.tensor([tokenizer.bos_id()] + tokenizer.encode(text), dtype=torch.int, device=device)
        torch.cuda.synchronize()
        prof = contextlib.nullcontext()
        if args.profile and i == args.num_samples - 1:
            prof = torch.profiler.profile()
        t0 = time.perf_counter()
        with prof:
            y = generate(model, prompt, args.max_new_tokens, decoder, temperature=args.temperature, top_k=args.top_k,
                         prefill=prefill)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        if i == -1:
            print(f"Graph capture + warm-up time: {t:.2f} seconds")
            continue
        if hasattr(prof, "export_chrome_trace"):
            prof.export_chrome_trace(f"{args.profile}.json")
        n_gen = y.size(0) - prompt.size(0)
        tps.append(n_gen / t)
        seqs.append(y.tolist())
        if tokenizer is not None:
            print(tokenizer.decode(y.tolist()))
        print(f"Time for inference {i + 1}: {t:.02f} sec total, {tps[-1]:.02f} tokens/sec")
        print(f"Bandwidth achieved: {model_size * tps[-1] / 1e9:.02f} GB/s (dense parameter bytes x tok/s, as the reference reports)")
    print("==========")
    if thresholds is not None and args.report_kept:
        report_kept_fractions(model, thresholds, prompt)
        if use_engine and args.max_new_tokens > 4:
            eng = decoder.model  # all seven projections, on the DECODE activations of the generated range
            kf = eng.mean_kept_fractions(prompt[-1:].clone(), prompt.numel(), min(args.max_new_tokens - 1, eng.max_seq - prompt.numel()), 3)
            print("achieved kept fraction (decode activations):", {k: round(v, 3) for k, v in kf.items()})
    mean = sum(tps) / max(1, len(tps))
    print(f"Average tokens/sec: {mean:.2f}")
    print(f"Memory used: {torch.cuda.max_memory_reserved() / 1e9:.02f} GB")
    pre_name = None if prefill is None else type(prefill).__name__ + (f":{prefill.used}" if getattr(prefill, "used", None) else "")
    return {"tokens_per_sec": tps, "mean_tokens_per_sec": mean, "thresholds": thresholds, "decoder": type(decoder).__name__,
            "sequences": seqs, "prefill": pre_name}


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="TEAL decode harness (MI355X / HIP)")
    p.add_argument("--prompt", type=str, default="Hello, my name is")
    p.add_argument("--interactive", action="store_true", help="ask for a prompt before every sample (needs a tokenizer: not with "
                   "--synthetic); chat checkpoints get the reference's [INST] wrapping (generate.py:440-446)")
    p.add_argument("--num_samples", type=int, default=5)
    p.add_argument("--max_new_tokens", type=int, default=200)
    p.add_argument("--top_k", type=int, default=200)
    p.add_argument("--temperature", type=float, default=0.8)
    p.add_argument("--checkpoint_path", type=Path, default=Path("checkpoints/meta-llama/Llama-2-7b-chat-hf/model.pth"))
    p.add_argument("--compile", action="store_true", help="capture the decode step into a hipGraph")
    p.add_argument("--compile_prefill", action="store_true", help="capture the prompt pass into a hipGraph too "
                   "(the reference's flag of the same name, generate.py:540); implied by --compile here")
    p.add_argument("--eager_prefill", action="store_true", help="with --compile: keep the prompt pass op by op (no prefill graph)")
    p.add_argument("--module_prefill", action="store_true", help="with --compile: the prompt pass through the patched modules (under a "
                   "hipGraph) also for prompts of up to 8 tokens, instead of the hand-fused HIP prompt pass")
    p.add_argument("--profile", type=Path, default=None)
    p.add_argument("--speculate_k", type=int, default=5, help="accepted for command-line compatibility; only read with a draft model")
    p.add_argument("--draft_checkpoint_path", type=Path, default=None, help="speculative decoding is outside this build (the "
                   "reference marks it untested with TEAL: README.md:111, generate.py:393): giving a draft model is an error")
    p.add_argument("--device", type=str, default=default_device)
    # monkeypatch (reference flags)
    p.add_argument("--hist_path", type=str, default=None)
    p.add_argument("--sparsity", type=float, default=0.0)
    # extensions
    p.add_argument("--precision", choices=["fp16", "bf16"], default="fp16")
    p.add_argument("--greedy_lookup", type=str, default=None, help="models/<name>/lookup directory (block-wise greedy sparsities)")
    p.add_argument("--synthetic", type=str, default=None, help="architecture name, e.g. 7B, llama-3-8b, 70B")
    p.add_argument("--n_layer", type=int, default=None, help="override the layer count (synthetic smoke runs)")
    p.add_argument("--dense", action="store_true", help="do not monkeypatch: dense baseline")
    p.add_argument("--report_kept", action="store_true", help="print the achieved kept fraction per projection")
    p.add_argument("--engine", action="store_true", help="fused HIP decode step (teal_amd/gpt_fast/engine.py); implied by --compile "
                   "when thresholds are installed")
    p.add_argument("--no_engine", action="store_true", help="with --compile: drive the patched model exactly as the reference "
                   "harness does (model(token, pos) -> torch sampler, captured in a hipGraph) instead of the device-resident "
                   "engine loop; single-token calls still run the fused HIP decode step (Transformer.fused_decode)")
    p.add_argument("--no_fused_decode", action="store_true", help="op-by-op module path (torch.ops.teal.* + eager glue) for "
                   "single-token calls: A/B against the fused decode step")
    return p


if __name__ == "__main__":
    res = main(build_parser().parse_args())
    print(json.dumps({"mean_tokens_per_sec": res["mean_tokens_per_sec"]}))
