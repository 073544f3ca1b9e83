#!/usr/bin/env python3
"""bench.py — decode tokens/sec (bs=1) of the TEAL activation-sparsity path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched through
torch.distributed.run with one rank per GPU.  Single-batch decode does not shard (BASELINE.json
north_star: "no RCCL"), so N > 1 runs N independent replicas ("replicas only"): the
only collective is the barrier / max-reduction of the timing, `value` = N*K tokens / max time.

A "step" = one decode token of Llama-2-7B (fp16, random-init weights at the exact shapes, synthetic
prompt, thresholds calibrated to the configured sparsity on the synthetic activations) through the
hipGraph-captured decode step.  One JSON line on rank 0, with:
  roofline      HBM roofline of the dominant kernel (the MLP gate/up sparse GEMV), achieved =
                algorithmic bytes / average launch duration measured live with HIP events;
  cpu_baseline  the CPU oracle port (oracle/, OpenMP) timed on this box's host cores on a bounded
                sample of the same workload (one layer's projections + lm_head, scaled to a token).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def measured_read_ceiling_gbs():
    """The best read rate a kernel of this repo sustains ON THIS BOX, measured in this run: a 1 GiB dense GEMV through
    teal_dense_gemv (8192 x 65536 fp16, every row kept, one launch incl. its boundary; median of 9 after 3 warm-up launches —
    6.34 TB/s on the boxes of rounds 2-4, which differ by ~1 %).  `roofline.frac_of_measured_ceiling` divides by THIS number."""
    from teal_amd.kernels import sparse_gemv as K
    Z, N = 8192, 65536
    w = torch.empty(Z, N, device="cuda", dtype=torch.float16).normal_(0, 0.02).T  # [N, Z], strides (1, N): the W^T image [Z][N]
    x = torch.randn(1, 1, Z, device="cuda", dtype=torch.float16)
    ts = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.dense_gemv(x, w)
        e1.record()
        e1.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e-3)
    del w
    torch.cuda.empty_cache()
    return (Z * N * 2 + Z * 2 + N * 2) / float(np.median(ts)) / 1e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="7B", help="architecture (BASELINE configs[1] = Llama-2-7B)")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--sparsity", type=float, default=0.5)
    ap.add_argument("--prompt_tokens", type=int, default=6, help="prefill length (the headline config uses the reference's 6-token prompt)")
    ap.add_argument("--weights", default="16bit", choices=["16bit", "int8", "int4"],
                    help="int8 = weight-only int8 projections + lm_head; int4 = group-quantised (g32) projections, 16-bit lm_head "
                         "(teal_amd/quantize.py); NOT the headline config")
    ap.add_argument("--pair", type=int, default=None, help="engine: 1/0 force the fused gate|up PAIR launch on/off (A/B)")
    ap.add_argument("--tuning", default="", help="lpr,waves,split,unroll override for every GEMV launch (A/B sweeps)")
    ap.add_argument("--att_split", type=int, default=0, help="override the engine's split-KV factor (A/B; 0 = automatic)")
    ap.add_argument("--mode", default="auto", choices=["auto", "engine", "dropin"],
                    help="engine = fused HIP decode step; dropin = reference-shaped torch modules + torch.ops.teal.*")
    ap.add_argument("--n_layer", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense comparator run")
    ap.add_argument("--no-reference-dense", action="store_true",
                    help="skip the second comparator (the un-patched gpt-fast model under a hipGraph: a second copy of the weights)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the rocprofv3 --pmc subprocess that measures roofline.traffic in this run (the committed pass is reported instead)")
    ap.add_argument("--no-context-sweep", action="store_true", help="skip the value_at_context runs (1000 and 3800 cache positions)")
    ap.add_argument("--block_size", type=int, default=0, help="override the architecture's context length (RoPE table / cache limit) "
                    "for long-context experiments; 0 = the reference's value (2048 for 7B)")
    ap.add_argument("--graph-tokens", type=int, default=1,
                    help="decode steps per hipGraph replay (token, position and RNG counter are device-resident, so a graph can span "
                         "several tokens; the timed region still runs exactly --steps steps, the remainder one token per replay)")
    ap.add_argument("--profile-markers", action="store_true",
                    help="bracket the timed region with one marker dispatch each (compact_kernel on 64 elements) so that "
                         "scripts/summarize_prof.py can restrict a rocprofv3 trace / counter pass to the timed hipGraph replays")
    return ap.parse_args()


def _dist_on():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def dist_setup(n):
    """one process per GPU (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE).  RCCL ("nccl") on a
    GPU box; gloo when no GPU is visible (the CPU test of this replica/timing logic).
    TEAL_BENCH_FORCE_DIST=nccl|gloo takes the world > 1 branch with ONE rank as well (process group with device_id, barrier,
    float64 MAX all-reduce on the device): every collective of the N > 1 path executes on the one leased GPU
    (tests/test_rccl_one_rank.py) before a multi-GPU node runs it."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    force = os.environ.get("TEAL_BENCH_FORCE_DIST", "")
    have_gpu = torch.cuda.is_available()
    ndev = torch.cuda.device_count() if have_gpu else 0
    if have_gpu:
        torch.cuda.set_device(local % ndev)
    if world > 1 or force:
        import socket
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_PORT" not in os.environ:  # (only without a launcher: the forced one-rank group)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if have_gpu and ndev >= world and force != "gloo":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local % ndev))
        else:  # no GPU (the CPU test), or more ranks than GPUs (replicas sharing a device: RCCL refuses that; the timing
               # barrier / max are all the collectives there are)
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def barrier(world):
    if world > 1 or _dist_on():
        import torch.distributed as dist
        dist.barrier()


_MARK = {}


def profile_marker():
    """one dispatch with a name nothing else in a decode step uses (compact_kernel): rocprofv3 traces of this command are cut
    at these dispatches (scripts/summarize_prof.py) — outside the timed region, between the synchronisations"""
    from teal_amd import _lib, runtime
    if not _MARK:
        _MARK["x"] = torch.zeros(64, device="cuda", dtype=torch.float16)
        _MARK["idx"] = torch.zeros(64, device="cuda", dtype=torch.int32)
        _MARK["n"] = torch.zeros(1, device="cuda", dtype=torch.int32)
    _lib.load().teal_compact(_MARK["x"].data_ptr(), 0.5, 64, 0, _MARK["idx"].data_ptr(), _MARK["n"].data_ptr(), runtime.stream_ptr())
    _sync()


def timed_decode(step_fn, steps, warmup, world, markers=False):
    """W untimed steps, then exactly K steps between barrier+synchronize on both sides."""
    run = getattr(step_fn, "run", None)  # engine stepper: run(n) = exactly n decode steps (--graph-tokens: several per replay)
    if run is not None:
        run(warmup)
    else:
        for _ in range(warmup):
            step_fn()
    _sync()
    if markers:
        profile_marker()
    barrier(world)
    _sync()
    t0 = time.perf_counter()
    if run is not None:
        run(steps)
    else:
        for _ in range(steps):
            step_fn()
    _sync()
    barrier(world)
    t = time.perf_counter() - t0
    if markers:
        profile_marker()
    if world > 1 or _dist_on():  # the job's time is the slowest replica's
        import torch.distributed as dist
        tt = torch.tensor([t], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    return t


def aggregate_tokens_per_sec(world, steps, t):
    """replicas only (no data-path collective): every rank decodes `steps` tokens of its own stream."""
    return world * steps / t


def make_stepper(model, a, dense=False):
    """Build the graphed decode step on `model`; returns (step_fn, info)."""
    from teal_amd.gpt_fast import generate as G
    dev = "cuda"
    info = {}
    if not dense:
        ths = G.apply_sparsity(model, sparsity=a.sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
        info["thresholds"] = ths
    prompt = torch.randint(0, model.config.vocab_size, (6,), device=dev, dtype=torch.int,
                           generator=torch.Generator(device=dev).manual_seed(7))
    total = 6 + a.warmup + a.steps + 8
    model.setup_caches(max_batch_size=1, max_seq_length=min(total, model.config.block_size))
    with torch.no_grad():
        logits = model(prompt.view(1, -1), torch.arange(0, 6, device=dev))
        tok = G.sample(logits, temperature=0.8, top_k=200)[0]
        dec = G.GraphedDecoder(model, True, 0.8, 200)
        dec.tok.copy_(tok.view(1, 1))
        dec.pos.fill_(6)
        dec.capture()
    one = torch.ones(1, dtype=torch.int, device=dev)

    def step():
        dec.graph.replay()
        dec.tok.copy_(dec.out_tok.view(1, 1))  # feed the sampled token back, advance the position
        dec.pos.add_(one)

    return step, info


def roofline_dominant_kernel(model, a):
    """MLP gate-projection sparse GEMV (the instantiation that also serves up and qkv): algorithmic
    bytes / average launch duration, HIP events around a hipGraph of one launch per layer (each layer's
    own w1: 32 x 90 MB of distinct weights, far beyond the 256 MB Infinity Cache)."""
    from teal_amd import _lib, runtime
    L = _lib.load()
    cfg = model.config
    Z, N = cfg.dim, cfg.intermediate_size
    dt = model.output.weight.dtype
    g = torch.Generator(device="cuda").manual_seed(11)
    x = (torch.rand(1, 1, Z, device="cuda", generator=g) - 0.5).to(dt)  # benchmark law: tau = s/2
    tau = a.sparsity / 2 if a.sparsity > 0 else -1.0
    nnz = int((x.float().abs() > tau).sum())
    ws = runtime.reserve_workspace(Z, N)
    y = torch.empty(N, device="cuda", dtype=dt)
    code = runtime.dtype_code(dt)
    weights = [l.feed_forward.w1.weight for l in model.layers]
    for w in weights:
        assert w.stride() == (1, N)

    def launch(i):
        rc = L.teal_sparse_gemv(x.data_ptr(), weights[i].data_ptr(), y.data_ptr(), tau, Z, N, code, ws.data_ptr(),
                                ws.numel() * 4, runtime.stream_ptr())
        assert rc == 0

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        launch(0)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with runtime.graph_capture(graph):
        for i in range(len(weights)):
            launch(i)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / len(weights))
    t = float(np.median(ts))
    algo = nnz * N * 2 + Z * 2 + N * 2
    cfgv = (ctypes.c_int * 5)()
    L.teal_get_config(Z, N, 1, cfgv)
    return {"bound": "hbm", "achieved": algo / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": algo / t / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "kernel": f"sparse_gemv_kernel<{cfgv[0]},{cfgv[1]},{cfgv[3]},{'bf16' if code else 'f16'}> (Z={Z}, N={N}, nnz={nnz})",
            "algorithmic_bytes": algo, "us_per_launch": t * 1e6,
            "timing": "HIP events around a hipGraph of one launch per layer (distinct weights); includes the same-stream launch boundary"}


def gateup_launch_bytes(nnz_gate, nnz_up, Z, N, nslabs, pair=False, wbytes=2, qbytes=0):
    """Bytes of ONE fused gate|up launch, SURVEY 8(d) to the letter: `algorithmic` = kept rows of both matrices
    (nnz x N x e each) + Z x 2 (the activation vector the GEMV consumes) + N_out x 2 (gate and up, or h = silu(gate) * up
    when the launch is paired) [+ the dense quantisation parameters of int8 / int4 images, which every launch reads whatever
    is kept].  What the fused producer reads to BUILD that vector (residual, fp32 slabs of the previous projection, norm
    weight) and the keep masks a paired launch emits are overhead and reported apart as `producer`.
    `kept_fraction` = kept rows / rows over both matrices, on the state the launch ran on."""
    n_out = N if pair else 2 * N
    algo = int((nnz_gate + nnz_up) * N * wbytes) + int(qbytes) + Z * 2 + n_out * 2
    producer = Z * 2 + nslabs * Z * 4 + Z * 2 + ((N // 8) if pair else 0)  # residual + fp32 slabs + norm weight (+ the keep masks out)
    return {"algorithmic": algo, "producer": producer, "kept_fraction": (nnz_gate + nnz_up) / (2.0 * Z)}


def roofline_fields(algorithmic_bytes_per_launch, us_per_launch):
    """achieved / frac of the roofline object: algorithmic bytes per launch / average launch duration / 8 TB/s"""
    achieved = algorithmic_bytes_per_launch / (us_per_launch * 1e-6) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}


def roofline_engine_gateup(eng, a):
    """Dominant kernel of the fused decode step: the MLP gate|up launch (RESID_NORM producer + two weight matrices, ~90 MB at
    50 %).  One launch per layer inside a hipGraph, each layer's own w1/w3 and thresholds (2.9 GB of distinct weights >> 256 MB
    Infinity Cache), timed with HIP events on the launch stream.  Algorithmic bytes: gateup_launch_bytes (kept rows
    of both matrices + Z x 2 + the output), counted on the state the graph runs on; the producer's inputs are `producer_bytes`."""
    from teal_amd import runtime
    from teal_amd.gpt_fast.engine import GemvIn, TEAL_IN_RESID_NORM
    m, cfg = eng.model, eng.cfg
    Z, N = cfg.dim, eng.inter
    ns = eng.n_wo.value
    B = eng.resid[1]
    # the producer's output, recomputed in torch to count kept rows per layer
    ssum = eng.handover_sum("wo")  # the fp32 hand-over of wo, added in slice order
    h = (B.float() + ssum.to(B.dtype).float()).to(B.dtype)
    hf = h.float()
    xn = (hf * torch.rsqrt(hf.pow(2).mean() + eng.eps)).to(B.dtype)
    total_bytes, producer_bytes, kept, launches = 0, 0, 0.0, []
    for i, st in enumerate(eng.stages):
        k4_in, k4_out = st[6], st[7]
        x = (xn * m.layers[i].ffn_norm.weight).float().abs()
        nnz_g = int((x > k4_out.tau[0]).sum())
        nnz_u = int((x > k4_out.tau[1]).sum())
        wbytes = 0.5 if eng.int4 else (1 if eng.int8 else 2)  # int8: + the two scale vectors; int4: + the group parameters
        qbytes = 2 * N * 2 if eng.int8 else (2 * (Z // m.layers[i].feed_forward.w1.groupsize) * N * 4 if eng.int4 else 0)
        b = gateup_launch_bytes(nnz_g, nnz_u, Z, N, ns, pair=eng.pair, wbytes=wbytes, qbytes=qbytes)
        total_bytes += b["algorithmic"]; producer_bytes += b["producer"]; kept += b["kept_fraction"]
        gin = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=B.data_ptr(), slabs=eng.s_wo.data_ptr(), nslabs=ns, slabs_interleaved=1,
                     norm_weight=m.layers[i].ffn_norm.weight.data_ptr(), eps=eng.eps, resid_out=None)
        launches.append((gin, k4_out))

    def run_all():
        eng._stream = runtime.stream_ptr()
        for gin, gout in launches:
            eng._gemv(gin, gout, Z)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run_all()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with runtime.graph_capture(graph):
        run_all()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.median(ts))
    n = len(launches)
    kname = eng.describe_launch(launches[0][0], launches[0][1], Z)  # the instantiation run_gemv launches (per call, no global)
    live = False
    traffic, tsrc = (None, "disabled (--no-live-traffic)") if (getattr(a, "no_live_traffic", False) or eng.int8 or eng.int4) else pmc_traffic_live(cfg, a, kname, pair=eng.pair)
    if traffic is not None:
        live = True
    else:
        why = tsrc
        traffic, tsrc = pmc_traffic(kname)  # the committed pass of this command (profiles/), flagged as such
        if tsrc:
            tsrc = f"{tsrc}; live pass: {why}"
    ceiling = measured_read_ceiling_gbs()
    out = roofline_fields(total_bytes / n, t / n * 1e6)
    out.update({"frac_of_measured_ceiling": out["achieved"] / ceiling,
            "measured_ceiling": ceiling, "measured_ceiling_how": "1 GiB dense GEMV (8192 x 65536 fp16) through teal_dense_gemv in this run, "
            "median of 9 launches incl. the launch boundary", "traffic": traffic, "traffic_source": tsrc,
            "traffic_measured_in_this_run": live,
            "kernel": kname + f" (fused RMSNorm -> mask -> gate|up GEMV{' -> silu*mul' if eng.pair else ''}, Z={Z}, N=2x{N}"
                      + ("; int4: algorithmic bytes count kept ROWS at half a byte per weight + the dense group parameters — the kernel "
                         "fetches row PAIRS, 1.5x those weight bytes at 50 %" if eng.int4 else "") + ")",
            "algorithmic_bytes": total_bytes / n, "algorithmic_bytes_formula": "sum over gate, up of nnz x N x 2  +  Z x 2  +  N_out x 2 (SURVEY 8(d))",
            "producer_bytes": producer_bytes / n, "producer_bytes_note": "what the fused producer reads instead of x (residual + fp32 slabs + norm "
            "weight; + the keep masks a paired launch emits): overhead, NOT in algorithmic_bytes / achieved / frac",
            "kept_fraction": kept / n, "kept_fraction_note": "kept rows / rows over both matrices, mean over layers, ON THE STATE THE TIMED GRAPH RUNS ON "
            "(the line's top-level kept_fraction is the mean over three decode positions)",
            "us_per_launch": t / n * 1e6, "launches_timed": n,
            "timing": "HIP events (launch stream) around a hipGraph of one launch per layer with that layer's weights; "
                      "per-launch time includes the same-stream launch boundary, like rocprofv3's per-dispatch duration"})
    return out


def _finite(o):
    """non-finite floats -> None, recursively (json.dumps(..., allow_nan=False) then never raises on a measurement artefact)"""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def floor_model(st_sparse, st_dense, n_layer):
    """Why the ratio is what it is, from numbers measured in this run (DecodeEngine.stage_times on the sparse and on the dense
    engine): each GEMV launch type fitted as t = fixed + bytes / stream_rate through its two measured points (kept rows at the
    configured sparsity, every row kept).  `fixed_us` is everything a launch pays whatever it streams — the launch boundary,
    entry skew, the producer's round trip, list build, the memory pipeline's fill and drain, the tail (phase-stamp breakdown:
    profiles/r04_layer_experiments.txt) — and is the same for the dense comparator, which is why the layer ratio sits near
    (F + D) / (F + D / 2) and not near 2."""
    gemv = ("qkv", "wo", "gate_up", "down")
    out = {"launch": {}, "unit": "us per launch (hipGraph of that stage over all layers, HIP events, incl. the launch boundary)"}
    F = 0.0
    for k in gemv:
        ts, td, bs, bd = st_sparse[k], st_dense[k], st_sparse["bytes"][k], st_dense["bytes"][k]
        if not (td > ts and bd > bs):
            # no slope through these two points (--sparsity 0, or timing noise on a narrow launch): no fit for this launch type —
            # its whole time counts as fixed, and the line stays valid JSON (no NaN tokens)
            rate, fixed = None, ts
        else:
            rate = (bd - bs) / (td - ts)   # bytes per us = MB/s
            fixed = ts - bs / rate
        F += fixed
        out["launch"][k] = {"us_sparse": round(ts, 2), "us_dense": round(td, 2), "MB_sparse": round(bs / 1e6, 2), "MB_dense": round(bd / 1e6, 2),
                            "stream_TBps": None if rate is None else round(rate / 1e6, 2), "fixed_us": round(fixed, 2),
                            "of_8TBps_sparse": round(bs / ts / 8e6, 3), "of_8TBps_dense": round(bd / td / 8e6, 3)}
    att_s, att_d = st_sparse["attn"], st_dense["attn"]
    Ds = sum(st_sparse[k] - out["launch"][k]["fixed_us"] for k in gemv)
    Dd = sum(st_dense[k] - out["launch"][k]["fixed_us"] for k in gemv)
    out["attention_us"] = {"sparse": round(att_s, 2), "dense": round(att_d, 2)}
    out["layer_us"] = {"sparse_measured": round(st_sparse["layer"], 2), "dense_measured": round(st_dense["layer"], 2),
                       "sparse_sum_of_launches": round(sum(st_sparse[k] for k in gemv) + att_s, 2),
                       "dense_sum_of_launches": round(sum(st_dense[k] for k in gemv) + att_d, 2)}
    out["fixed_us_per_layer"] = round(F, 2)
    out["streaming_us_per_layer"] = {"sparse": round(Ds, 2), "dense": round(Dd, 2)}
    out["layer_ratio"] = {"measured": round(st_dense["layer"] / st_sparse["layer"], 3),
                          "model (fixed + attention + streaming)": round((F + att_d + Dd) / (F + att_s + Ds), 3),
                          "if the fixed per-launch cost and the attention launch were free": round(Dd / Ds, 3) if Ds > 0 else None}
    out["note"] = ("fixed_us = launch boundary + entry + producer + list + memory-pipeline fill/drain + tail, paid per launch by sparse and "
                   "dense alike; stream_TBps = the slope between the two measured points of a launch type")
    return out


def pmc_traffic_live(cfg, a, kernel_desc, pair=False, timeout_s=150):
    """HBM read bytes per launch of the dominant kernel, measured IN THIS RUN: scripts/pmc_gateup.py (the same launch on its
    own: same entry point, geometry and kept fraction) as a subprocess under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`
    (counters in a pass of their own, kernel trace only), FETCH_SIZE KiB x 2 as MI355X_MICROARCH.md prescribes for gfx950
    (calibrated on this access pattern: profiles/r04_bench_summary.txt, FETCH x 2 / algorithmic = 1.01-1.02 on four launches).
    Returns (bytes, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    # this process is itself being profiled (rocprofv3 / rocprofiler-sdk tool preloaded): no nested counter pass
    if any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "bench.py is running under a profiler: no nested counter pass"
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="teal_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [prof, "--kernel-trace", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", out, "-o", "gateup", "--", sys.executable,
           os.path.join(ROOT, "scripts", "pmc_gateup.py"), "--dim", str(cfg.dim), "--inter", str(cfg.intermediate_size),
           "--sparsity", str(a.sparsity), "--dtype", a.precision, "--pair", str(int(pair))]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd="/tmp")
        if r.returncode != 0:
            return None, f"rocprofv3 pass failed (rc {r.returncode})"
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        info = json.loads(line[-1]) if line else {}
        want = info.get("kernel", kernel_desc).split(" grid")[0].replace(" ", "")
        vals = []
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == "FETCH_SIZE" and "gemv_fast_kernel" in row["Kernel_Name"]:
                    m = re.search(r"gemv_fast_kernel<([^>]*)>", row["Kernel_Name"])
                    if m and ("gemv_fast_kernel<" + m.group(1).replace(" ", "") + ">") == want:
                        vals.append(float(row["Counter_Value"]))
        if len(vals) < 4:
            return None, "no counter rows for the dominant kernel"
        vals = vals[len(vals) // 4:]  # drop the first passes (cold TLBs / first touch)
        return (sum(vals) / len(vals)) * 1024 * 2, {
            "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE over scripts/pmc_gateup.py, a subprocess of this run (FETCH_SIZE KiB x 2, per dispatch)",
            "dispatches": len(vals), "kept_fraction": info.get("kept_fraction"), "algorithmic_bytes_of_that_launch": info.get("algorithmic_bytes")}
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
        return None, f"counter pass not available: {type(e).__name__}"
    finally:
        shutil.rmtree(out, ignore_errors=True)


def pmc_traffic(kernel_desc):
    """HBM read bytes per launch of the dominant kernel.  PMC counters cannot be sampled from inside the process: the
    value comes from the COMMITTED rocprofv3 --pmc FETCH_SIZE pass of this same command (newest
    profiles/*pmc_fetch_by_kernel.csv) and is reported only if that profile names the kernel this run launched — it is
    flagged `traffic_measured_in_this_run: false`.  FETCH_SIZE is in KiB and, on gfx950, counts 128-byte requests as 64
    bytes for wide coalesced reads: doubled as MI355X_MICROARCH.md (HBM section) prescribes."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_fetch_by_kernel.csv")))
    if not files:
        return None, None
    # "gemv_fast_kernel<...> grid (172,1) x 1024": the same instantiation serves launches with different grids (gate|up and
    # lm_head), so the profile is keyed by (instantiation, workgroups)
    short = kernel_desc.split(" grid")[0].replace(" ", "")
    m = re.search(r"grid \((\d+),(\d+)\)", kernel_desc) or re.search(r"grid (\d+) x", kernel_desc)
    wgs = None
    if m:
        wgs = int(m.group(1)) * (int(m.group(2)) if m.lastindex and m.lastindex > 1 else 1)
    for r in csv.DictReader(open(files[-1])):
        if short and short == r["kernel"].replace(" ", "") and r["counter"] == "FETCH_SIZE" and (
                "workgroups" not in r or wgs is None or int(r["workgroups"]) == wgs):
            return float(r["avg_value"]) * 1024 * 2, os.path.relpath(files[-1], ROOT) + " (FETCH_SIZE KiB x 2, per dispatch)"
    return None, None


def cpu_baseline(model, a, budget_s=14.0, n_layers_sampled=4):
    """Oracle port (fp32 accumulate, OpenMP; resident-matrix form, oracle/teal_oracle.c: teal_oracle_mat_*) on the five sparse
    projections of `n_layers_sampled` layers + the dense lm_head — ~1.9 GB of distinct weights per pass, so that the pass
    streams from DRAM and not from a large last-level cache (one layer alone, 0.4 GB, fits the L3 of this class of host and
    measured 440 GB/s "dense") — inputs U(-.5,.5) with tau = s/2 (kept fraction 1-s), scaled to a whole token."""
    # the oracle's OpenMP runtime (the system libgomp, not torch's bundled copy) reads these when it first starts: threads
    # stay where they first touched their share of every matrix
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "threads")
    from oracle import teal_oracle as O
    cfg = model.config
    dt = model.output.weight.dtype
    code = 0 if dt == torch.float16 else 1
    tau = a.sparsity / 2 if a.sparsity > 0 else -1.0
    nl = min(n_layers_sampled, cfg.n_layer)

    def host_bits(w):  # [N, Z] column-major -> W^T [Z][N] bits
        N, Z = w.shape
        return w.detach().T.contiguous().cpu().view(torch.int16).numpy().view(np.uint16).reshape(-1), Z, N

    host = []
    for layer in model.layers[:nl]:
        mats = {"qkv": layer.attention.wqkv.weight, "o": layer.attention.wo.weight, "gate": layer.feed_forward.w1.weight,
                "up": layer.feed_forward.w3.weight, "down": layer.feed_forward.w2.weight}
        host.append({k: host_bits(v if v.stride(0) == 1 else v.T.contiguous().T) for k, v in mats.items()})
    lm = model.output.weight.detach().T.contiguous().cpu().view(torch.int16).numpy().view(np.uint16).reshape(-1)
    xs = {k: O.hash_uniform(Z, 100 + Z, 1.0, code) for k, (_, Z, _) in host[0].items()}
    x_lm = O.hash_uniform(cfg.dim, 99, 1.0, code)

    def prepare():
        return [{k: O.Mat(wb, Z, N, code) for k, (wb, Z, N) in h.items()} for h in host], O.Mat(lm, cfg.dim, cfg.vocab_size, code)

    def release(layers_r, lm_r):
        for d in layers_r:
            for m_ in d.values():
                m_.close()
        lm_r.close()

    def one_pass(layers_r, lm_r, t):  # -> (seconds in the layers, seconds in the lm_head)
        t0 = time.perf_counter()
        for d in layers_r:
            for k, m_ in d.items():
                m_.gemv(xs[k], t)
        t1 = time.perf_counter()
        lm_r.gemv(x_lm, -1.0)
        return t1 - t0, time.perf_counter() - t1

    # OpenMP width: the fastest of a few candidates on one DRAM-bound pass (a Mat is bound to the width it was created under)
    ncpu = os.cpu_count() or 1
    best, best_t = None, float("inf")
    for n in sorted({c for c in (16, 32, 64, 128, 192, 256) if c <= ncpu} | {min(ncpu, 8)}):
        O.set_threads(n)
        lr, lmr = prepare()
        one_pass(lr, lmr, tau)
        tl, tm = one_pass(lr, lmr, tau)
        if tl + tm < best_t:
            if best is not None:
                release(*best[1])
            best, best_t = (n, (lr, lmr)), tl + tm
        else:
            release(lr, lmr)
    used_threads, (layers_r, lm_r) = best
    O.set_threads(used_threads)
    acc = {"sparse": [0.0, 0.0, 0], "dense": [0.0, 0.0, 0]}
    for leg, t, share in (("sparse", tau, 0.6), ("dense", -1.0, 0.4)):
        one_pass(layers_r, lm_r, t)
        t_end = time.perf_counter() + budget_s * share
        while time.perf_counter() < t_end and acc[leg][2] < 200:
            tl, tm = one_pass(layers_r, lm_r, t)
            acc[leg][0] += tl; acc[leg][1] += tm; acc[leg][2] += 1
    reps, r3 = acc["sparse"][2], acc["dense"][2]
    t_layer = acc["sparse"][0] / reps / nl
    t_lm = (acc["sparse"][1] + acc["dense"][1]) / (reps + r3)
    t_layer_dense = acc["dense"][0] / r3 / nl
    t_token = cfg.n_layer * t_layer + t_lm
    t_token_dense = cfg.n_layer * t_layer_dense + t_lm
    kept_bytes = sum(int((np.abs(O.from_bits(xs[k], code)) > tau).sum()) * N * 2 for k, (_, Z, N) in host[0].items())
    dense_bytes = sum(Z * N * 2 for _, Z, N in host[0].values())
    release(layers_r, lm_r)
    host_ceiling = O.host_read_gbs(2 << 30, 3)  # a plain parallel sum over 2 GiB at the same OpenMP width
    return {"value": 1.0 / t_token, "unit": "tokens/s", "cores": used_threads, "kind": "port", "host_read_ceiling_gbs": host_ceiling,
            "dense_value": 1.0 / t_token_dense, "ms_per_layer_dense": t_layer_dense * 1e3,
            "host_gbs_sparse": kept_bytes / t_layer / 1e9, "host_gbs_dense": dense_bytes / t_layer_dense / 1e9,
            "host_gbs_lm_head": cfg.dim * cfg.vocab_size * 2 / t_lm / 1e9,
            "sample": f"oracle/teal_oracle.c resident-matrix port (matrices prepared once: tile-major, first-touch placed per "
                      f"thread, no per-call allocation): {nl} of {cfg.n_layer} layers (qkv, o, gate, up, down sparse GEMVs at kept "
                      f"fraction {1 - a.sparsity:.2f}) + the dense lm_head per pass (~{(nl * dense_bytes + cfg.dim * cfg.vocab_size * 2) / 1e9:.1f} GB of "
                      f"distinct weights, beyond the host's last-level cache) x {reps} passes sparse / {r3} dense, scaled to one token "
                      f"({cfg.n_layer} layers + lm_head); GEMVs only (no attention/norms), so it flatters the CPU",
            "ms_per_layer": t_layer * 1e3, "ms_lm_head": t_lm * 1e3}


def main():
    a = parse()
    if a.tuning:  # a forced launch geometry is a switch of the diagnostics build (libteal_hip_diag.so): the whole run goes through it
        os.environ["TEAL_LIB_FLAVOR"] = "diag"
    rank, world, local = dist_setup(a.gpus)
    from teal_amd import runtime
    from teal_amd.gpt_fast import generate as G
    runtime.init()
    from teal_amd import _lib
    if a.tuning:
        assert _lib.load().teal_set_tuning(*[int(v) for v in a.tuning.split(",")]) == 0
    if a.pair is not None:
        a.pair = bool(a.pair)
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.precision]
    torch.manual_seed(1234)
    mode = a.mode
    if mode == "auto":
        try:
            from teal_amd.gpt_fast import engine  # noqa: F401
            mode = "engine"
        except ImportError:
            mode = "dropin"

    model = G.build_synthetic_model(a.model, "cuda", dt, n_layer=a.n_layer)
    if a.block_size:
        model.config.block_size = a.block_size
    if a.prompt_tokens + 8 > model.config.block_size:
        raise SystemExit(f"--prompt_tokens {a.prompt_tokens} does not fit the model's context (block_size "
                         f"{model.config.block_size}); pass --block_size for a long-context experiment")
    if a.weights == "int8":
        from teal_amd.quantize import quantize_model_int8
        quantize_model_int8(model)
        torch.cuda.empty_cache()
        assert mode == "engine"
    if a.weights == "int4":
        from teal_amd.quantize import quantize_model_int4
        quantize_model_int4(model, 32)
        torch.cuda.empty_cache()
        assert mode == "engine"
    cfg = model.config
    extra = {}
    if mode == "engine":
        from teal_amd.gpt_fast.engine import make_engine_stepper
        step, info = make_engine_stepper(model, a)
    else:
        step, info = make_stepper(model, a)
    t = timed_decode(step, a.steps, a.warmup, world, markers=a.profile_markers)
    tps = aggregate_tokens_per_sec(world, a.steps, t)
    mname = {"7B": "Llama-2-7B", "13B": "Llama-2-13B", "70B": "Llama-2-70B", "llama-3-8b": "Llama-3-8B"}.get(a.model, a.model)
    out = {"metric": f"decode tokens/sec (bs=1), {mname} {a.precision} @{a.sparsity:.0%} activation sparsity", "value": tps, "unit": "tokens/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": t / a.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if dt == torch.float16 else "bf16",
           "data": "synthetic (random-init weights at exact shapes, random token ids, thresholds calibrated to the kept fraction)",
           "config": {"workload": f"Llama-2-{a.model} bs=1 decode, uniform {a.sparsity:.0%} sparsity, hipGraph-captured step"
                      if a.model == "7B" else f"{a.model} bs=1 decode, uniform {a.sparsity:.0%} sparsity",
                      "mode": mode, "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
                      "prompt_tokens": a.prompt_tokens, "context_positions": f"{a.prompt_tokens}..{a.prompt_tokens + a.warmup + a.steps}"}}
    if a.weights == "int8":
        out["metric"] = out["metric"].replace(a.precision, f"int8-weight/{a.precision}-activation")
        out["dtype"] = out["dtype"] + " activations, int8 weights (per-channel scales)"
        out["config"]["workload"] += ", int8 weight-only"
    if a.weights == "int4":
        out["metric"] = out["metric"].replace(a.precision, f"int4-g32-weight/{a.precision}-activation")
        out["dtype"] = out["dtype"] + " activations, int4 group-quantised weights (g32), 16-bit lm_head"
        out["config"]["workload"] += ", int4 group-quantised projections (g32), 16-bit lm_head"
    if _dist_on():  # which collectives bracketed the timed region (N > 1, or the forced one-rank group)
        import torch.distributed as dist
        out["config"]["process_group"] = f"{dist.get_backend()}, world {world}: barrier + max over ranks of the timing"
    out.update(info.get("report", {}))
    if info.get("graph_tokens", 1) > 1:
        out["config"]["graph_tokens"] = info["graph_tokens"]  # decode steps per hipGraph replay (device-resident loop state)
    if rank == 0 and mode == "engine":
        eng = info["engine"]
        # achieved kept fraction of all seven projections on the DECODE activations of one more step
        out["kept_fraction"] = {k: round(v, 4) for k, v in
                                eng.mean_kept_fractions(info["first_token"], info["pos0"], info["span"], 3).items()}
        out["kept_fraction_note"] = "mean over layers and over 3 decode positions (first, middle, last) of the timed range"
        # the reference's tokens/sec definition (gpt-fast/generate.py:487-497): generated tokens / (prefill + decode) for
        # one sample of max_new_tokens = 200 after a prompt of `prompt_tokens`
        t_pf = info.get("prefill_s")
        if t_pf:
            # (max_new_tokens = 200 is the prompt pass — which yields the first new token — plus 199 decode steps: generate.py:380-406)
            out["tokens_per_sec_reference_definition"] = 200.0 / (t_pf + 199.0 * t / a.steps)
            out["prefill_ms"] = t_pf * 1e3
            out["prefill_path"] = {"hip": "hand-fused HIP prompt pass (teal_amd/gpt_fast/prefill.py), hipGraph replay",
                                   "fallback": "module path"}.get(info.get("prefill_path"), info.get("prefill_path"))
    st_sparse = None
    if rank == 0 and world == 1:
        out["roofline"] = roofline_engine_gateup(info["engine"], a) if mode == "engine" else roofline_dominant_kernel(model, a)
        if mode == "engine" and not a.no_dense and info["engine"].att_fused_merge and not info["engine"].int4:
            st_sparse = info["engine"].stage_times()  # (int4: kept PAIRS + dense group parameters — another byte model)
        if not a.no_dense:
            # dense comparator on the same harness: same kernels with every row kept (threshold < 0)
            a_d = argparse.Namespace(**vars(a))
            a_d.sparsity = 0.0
            dmodel = model
            if mode == "engine":
                from teal_amd.gpt_fast.engine import make_engine_stepper
                dstep, dinfo = make_engine_stepper(dmodel, a_d)
            else:
                dstep, dinfo = make_stepper(dmodel, a_d)
            td = timed_decode(dstep, max(20, a.steps // 2), max(5, a.warmup // 2), 1)
            if st_sparse is not None:
                out["floor_model"] = floor_model(st_sparse, dinfo["engine"].stage_times(), cfg.n_layer)
            dense_tps = max(20, a.steps // 2) / td
            out["dense_tokens_per_sec"] = dense_tps
            out["speedup_vs_dense"] = tps / dense_tps
            out["dense_comparator"] = ("the same fused engine with every row kept (thresholds < 0: same kernels, same launches) — "
                                       "the strongest dense fp16 path on this box; the reference's own dense path follows")
            if mode == "engine" and a.weights == "16bit" and not a.no_reference_dense:
                # the reference's dense path as it exists here: the un-patched gpt-fast model (torch.matmul / hipBLASLt,
                # eager glue; no Inductor on this image) under the same hipGraph capture
                rmodel = G.build_synthetic_model(a.model, "cuda", dt, n_layer=a.n_layer)
                rstep, _ = make_stepper(rmodel, a_d, dense=True)
                nref = max(10, a.steps // 8)
                tr = timed_decode(rstep, nref, 3, 1)
                out["reference_dense_path_tokens_per_sec"] = nref / tr
                out["speedup_vs_reference_dense_path"] = tps / (nref / tr)
                del rmodel, rstep
                torch.cuda.empty_cache()
        if mode == "engine" and not a.no_context_sweep and a.prompt_tokens < 500:
            # The headline decodes at the reference's default prompt (6 tokens): cache positions 6..~230, the short-context
            # best case of the attention launch.  Same model, same thresholds law, longer contexts (a prefill of that many
            # random tokens through the module path, thresholds re-taken on the timed decode positions):
            from teal_amd.gpt_fast.engine import make_engine_stepper
            out["value_at_context"] = {}
            for ctx in (1000, 3800):
                a_c = argparse.Namespace(**vars(a))
                a_c.prompt_tokens, a_c.steps, a_c.warmup = ctx, min(a.steps, 60), min(a.warmup, 10)
                model.config.block_size = max(model.config.block_size, 4096 if ctx > 1900 else 2048)
                cstep, cinfo = make_engine_stepper(model, a_c, ths=info["thresholds"])
                tc = timed_decode(cstep, a_c.steps, a_c.warmup, 1)
                out["value_at_context"][str(ctx)] = a_c.steps / tc
                # the kept fractions ON THOSE positions (the o projection's input shrinks with the context on synthetic weights;
                # thresholds were re-taken on the timed range of this context)
                out.setdefault("kept_fraction_at_context", {})[str(ctx)] = {
                    k: round(v, 4) for k, v in cinfo["engine"].mean_kept_fractions(cinfo["first_token"], cinfo["pos0"], cinfo["span"], 2).items()}
                del cstep, cinfo
                torch.cuda.empty_cache()
            out["value_at_context_note"] = ("tokens/s of the same decode step with ~1000 / ~3800 cached positions (block_size raised to "
                                            "4096 for the latter); `value` is the reference's default 6-token prompt")
        if not a.no_cpu_baseline and a.weights == "16bit":
            out["cpu_baseline"] = cpu_baseline(model, a)
    elif world > 1:
        # N > 1: the timing's collectives are done — the ranks leave the group TOGETHER (barrier first), then rank 0 alone measures
        # the dominant launch on its own GPU (a per-GPU property: replicas do not change it).  No dense leg, no CPU baseline (N = 1
        # only, by the bench contract).
        import torch.distributed as dist
        barrier(world)
        dist.destroy_process_group()
        if rank == 0:
            out["roofline"] = roofline_engine_gateup(info["engine"], a) if mode == "engine" else roofline_dominant_kernel(model, a)
            out["roofline"]["note_n_gpus"] = f"measured on rank 0's GPU after the timed region (each of the {world} replicas runs the same launch)"
            out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(_finite(out), allow_nan=False))  # strict JSON: a non-finite number would print as a bare NaN token
    if _dist_on():
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
