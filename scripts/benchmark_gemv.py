#!/usr/bin/env python3
"""Kernel micro-benchmark — counterpart of the reference's scripts/benchmark_gemv.py:186-258.

Same experiment: a single fp16 GEMV, x = U(-.5,.5) [1,1,Z], W = U(-.5,.5) [N,Z] column-major,
threshold = s/2 so that exactly a fraction s of the activations is dropped (benchmark_gemv.py:164-168,
199-203), swept over sparsity; providers:
    teal_hip      this repo's sparse GEMV (C ABI)            <- 'zeal'  (splitk_sparse_gemv, :211)
    dense_torch   x @ W_T (hipBLASLt/rocBLAS via torch)       <- 'dense' (:206)
    dense_hip     this repo's kernel with every row kept
    theoretical   dense_torch * (1 - s)                       <- 'theoretical optimal' (:218-223)
    cpu_port      oracle fp32 port on the host cores (a few sparsities only; test infrastructure, used
                  here as the same-box CPU baseline)
    deja_vu       the Deja Vu gather kernel's METHOD restated in HIP (precomputed flags, fp32 atomics into a zeroed
                  output, three launches; teal_cmp_flag_gemv)          <- 'deja vu' (:32-107, 170-172, 214)
Timing: median / p20 / p80 over hipGraph replays of 32 back-to-back launches ROTATING over > 1 GB of
distinct weight buffers (the reference's do_bench re-reads one 117 MB matrix, which on MI355X would sit
in the 256 MB Infinity Cache).  Writes CSV like the reference (ms per call) + GB/s of algorithmic bytes.
"""
import argparse
import csv
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teal_amd import _lib, runtime  # noqa: E402


def graph_times(fn, nlaunch, reps):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nlaunch):
            fn(i)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / nlaunch)  # ms per call
    ts.sort()
    return ts[len(ts) // 2], ts[int(len(ts) * 0.2)], ts[int(len(ts) * 0.8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in_size", type=int, default=4096)
    ap.add_argument("--out_size", type=int, default=14336)  # the reference's shape (:195); BASELINE config 1 names 4096 too
    ap.add_argument("--step", type=float, default=0.05)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "benchmark_gemv"))
    ap.add_argument("--pad", type=int, default=64, help="row padding of W^T (0 = the reference's exact layout)")
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    L = _lib.load()
    runtime.init()
    Z, N = a.in_size, a.out_size
    dt = torch.float16
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.rand(1, 1, Z, device="cuda", generator=g) - 0.5).to(dt)
    ld = N + a.pad
    nbuf = int(1.1e9 / (Z * ld * 2)) + 1
    bufs = []
    for _ in range(nbuf):
        b = torch.zeros(Z, ld, device="cuda", dtype=dt)
        b[:, :N] = (torch.rand(Z, N, device="cuda", generator=g) - 0.5).to(dt)
        bufs.append(b)
    dense_w = [b[:, :N] for b in bufs]  # W_T views for x @ W_T
    ws = runtime.reserve_workspace(Z, N)
    y = torch.empty(N, device="cuda", dtype=dt)
    st = runtime.stream_ptr

    def dense(i):
        torch.matmul(x.view(1, -1), dense_w[i % nbuf])

    d_ms = graph_times(dense, 32, 9)
    y32 = torch.zeros(N, device="cuda", dtype=torch.float32)
    flags = torch.zeros(Z, device="cuda", dtype=torch.uint8)
    rows = []
    levels = [round(i * a.step, 4) for i in range(int(round(1 / a.step)))] + [0.99]
    O = None
    if a.cpu:
        from oracle import teal_oracle as Omod
        O = Omod
        wb_host = bufs[0][:, :N].contiguous().cpu().view(torch.int16).numpy().view(np.uint16).reshape(-1)
        xb_host = x.view(-1).cpu().view(torch.int16).numpy().view(np.uint16)
        cpu_mat = O.Mat(wb_host, Z, N, 0)  # resident-matrix port (prepared once; one 33-117 MB matrix: last-level-cache resident)
    for s_ in levels:
        tau = s_ / 2 if s_ > 0 else -1.0
        nnz = int((x.float().abs() > tau).sum())

        def sparse(i, tau=tau):
            rc = L.teal_sparse_qkv_gemv_ld(x.data_ptr(), bufs[i % nbuf].data_ptr(), ld, y.data_ptr(), tau, tau, tau, Z, N, N, 0, 0,
                                           ws.data_ptr(), ws.numel() * 4, st())
            assert rc == 0

        ms = graph_times(sparse, 32, 9)

        def sparse_resident(i, tau=tau):  # the SAME 33-117 MB matrix every launch: served by the 256 MB Infinity Cache
            sparse(0, tau)

        ms_res = graph_times(sparse_resident, 32, 9)
        def dejavu(i, tau=tau):
            rc = L.teal_cmp_flag_gemv(x.data_ptr(), bufs[i % nbuf].data_ptr(), ld, y32.data_ptr(), flags.data_ptr(), tau, Z, N, 0, st())
            assert rc == 0

        ms_dv = graph_times(dejavu, 32, 9)
        if s_ in (0.0, 0.5):  # the comparator computes the same masked GEMV
            sparse(0); dejavu(0); torch.cuda.synchronize()
            assert torch.allclose(y32, y.float(), atol=2e-2, rtol=2e-2), float((y32 - y.float()).abs().max())
        algo = nnz * N * 2 + Z * 2 + N * 2
        row = {"sparsity_level": s_, "nnz": nnz, "TEAL_HIP": ms[0], "TEAL_HIP_min": ms[1], "TEAL_HIP_max": ms[2],
               "Dense": d_ms[0], "Dense_min": d_ms[1], "Dense_max": d_ms[2], "Theoretical Optimal": d_ms[0] * (1 - s_),
               "TEAL_HIP_GBps": algo / (ms[0] * 1e-3) / 1e9, "speedup_vs_dense": d_ms[0] / ms[0],
               "Deja Vu": ms_dv[0], "Deja Vu_min": ms_dv[1], "Deja Vu_max": ms_dv[2],
               "TEAL_HIP_cache_resident": ms_res[0]}  # labelled: NOT an HBM number
        if O is not None and abs(s_ * 20 - round(s_ * 20)) < 1e-9 and s_ in (0.0, 0.25, 0.5, 0.75):
            cpu_mat.gemv(xb_host, tau)
            t0 = time.perf_counter()
            for _ in range(20):
                cpu_mat.gemv(xb_host, tau)
            row["CPU_port_ms"] = (time.perf_counter() - t0) / 20 * 1e3
            row["CPU_threads"] = O.num_threads()
        rows.append(row)
        print(f"s={s_:.2f} nnz={nnz:5d}  hip {ms[0]*1e3:7.2f} us ({row['TEAL_HIP_GBps']:7.1f} GB/s)  deja vu {ms_dv[0]*1e3:7.2f} us  dense {d_ms[0]*1e3:7.2f} us  "
              f"speed-up {row['speedup_vs_dense']:.2f}x  [cache-resident {ms_res[0]*1e3:6.2f} us]" + (f"  cpu {row['CPU_port_ms']:.2f} ms" if "CPU_port_ms" in row else ""))
    path = os.path.join(a.out, f"Kernel Plot (MI355X) ({Z}x{N}).csv")
    keys = sorted({k for r in rows for k in r}, key=lambda k: (k != "sparsity_level", k))
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keys)
        w.writeheader()
        w.writerows(rows)
    print(f"Results saved to {path}")


if __name__ == "__main__":
    main()
