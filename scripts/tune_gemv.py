#!/usr/bin/env python3
"""Launch-geometry sweep of the sparse GEMV on the GPU box (run through gpurun).

For every (shape, sparsity) it times every (lanes_per_row, waves, split, unroll) variant with the
weights ROTATING over > 1 GB of distinct buffers (the 256 MB Infinity Cache would otherwise serve a
33 MB matrix), inside a hipGraph of back-to-back launches (so the number is kernel + one
same-stream boundary, what a decode step pays).  Writes JSON lines to gpurun_out/tune_*.jsonl and
prints the best variants.  Not part of the product path; a benchmark utility.
"""
import argparse
import ctypes
import itertools
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TEAL_LIB_FLAVOR", "diag")  # the tuning / phase-stamp switches exist in libteal_hip_diag.so only
from teal_amd import _lib, runtime  # noqa: E402


def algo_bytes(nnz, N, Z, nmat=1):
    return nmat * nnz * N * 2 + Z * 2 + nmat * N * 2


def make_case(Z, N, sparsity, kind, dtype, total_gb):
    """x ~ U(-0.5, 0.5), tau = s/2 (scripts/benchmark_gemv.py:164-168, 199-203)."""
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.rand(1, 1, Z, device=dev, generator=g) - 0.5).to(dtype)
    tau = sparsity / 2 if sparsity > 0 else -1.0
    nmat = 2 if kind == "gateup" else 1
    per = Z * N * 2 * nmat
    nbuf = max(2, int(total_gb * 1e9 / per) + 1)
    bufs = []
    for _ in range(nbuf):
        bufs.append([(torch.rand(Z, N, device=dev, generator=g) - 0.5).to(dtype) for _ in range(nmat)])
    nnz = int((x.float().abs() > tau).sum())
    return x, tau, bufs, nnz, nmat


def time_graph(fn_launch, nbuf, launches, reps, warm=3):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn_launch(0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for i in range(launches):
            fn_launch(i % nbuf)
    for _ in range(warm):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / launches)  # us per launch
    ts.sort()
    return ts[len(ts) // 2], ts[int(len(ts) * 0.2)], ts[int(len(ts) * 0.8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tune_gemv.jsonl"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--total-gb", type=float, default=1.1)
    ap.add_argument("--launches", type=int, default=48)
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    L = _lib.load()
    ncu = runtime.init()
    print(f"CUs: {ncu}  device: {torch.cuda.get_device_name(0)}")
    dtype = torch.float16
    cases = [
        ("wo_7b", 4096, 4096, 0.5, "gemv"),
        ("gate_7b", 4096, 11008, 0.5, "gemv"),
        ("gateup_7b", 4096, 11008, 0.5, "gateup"),
        ("down_7b", 11008, 4096, 0.5, "gemv"),
        ("qkv_7b", 4096, 12288, 0.5, "qkv"),
        ("bench_14336", 4096, 14336, 0.5, "gemv"),
        ("lmhead_7b", 4096, 32000, 0.0, "gemv"),
        ("wo_7b_dense", 4096, 4096, 0.0, "gemv"),
        ("gateup_7b_dense", 4096, 11008, 0.0, "gateup"),
        ("down_7b_dense", 11008, 4096, 0.0, "gemv"),
    ]
    if a.shapes:
        cases = [c for c in cases if c[0] in a.shapes.split(",")]
    lprs, waves, unrolls = (8, 16, 32, 64), (16,), (4,)  # 8-wave / unroll-8 variants were sweep-only (round 1) and are no longer built
    if a.quick:
        lprs, waves, unrolls = (8, 32), (16,), (4,)

    # calibrate: device-to-device copy bandwidth (read + write bytes)
    src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    e1.synchronize()
    copy_tbs = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(f"D2D copy: {copy_tbs:.2f} TB/s (read+write)")
    del src, dst

    out = open(a.out, "w")
    out.write(json.dumps({"kind": "meta", "ncu": ncu, "copy_tbs": copy_tbs, "device": torch.cuda.get_device_name(0)}) + "\n")
    t_start = time.time()
    for tag, Z, N, s, kind in cases:
        x, tau, bufs, nnz, nmat = make_case(Z, N, s, kind, dtype, a.total_gb)
        nbuf = len(bufs)
        ws = runtime.reserve_workspace(Z, N)
        y = torch.empty(N * nmat, device="cuda", dtype=dtype)
        ab = algo_bytes(nnz, N, Z, nmat)
        st = lambda: runtime.stream_ptr()  # noqa: E731

        def launch(i, kind=kind, bufs=bufs, x=x, y=y, ws=ws, Z=Z, N=N, tau=tau):
            if kind == "gemv":
                rc = L.teal_sparse_gemv(x.data_ptr(), bufs[i][0].data_ptr(), y.data_ptr(), tau, Z, N, 0, ws.data_ptr(), ws.numel() * 4, st())
            elif kind == "qkv":
                rc = L.teal_sparse_qkv_gemv(x.data_ptr(), bufs[i][0].data_ptr(), y.data_ptr(), tau, tau, tau, Z, N, N - 2 * 4096, 4096, 0,
                                            ws.data_ptr(), ws.numel() * 4, st())
            else:
                rc = L.teal_sparse_gateup_silu(x.data_ptr(), bufs[i][0].data_ptr(), bufs[i][1].data_ptr(), y.data_ptr(), tau, tau, Z, N, 0,
                                               ws.data_ptr(), ws.numel() * 4, st())
            assert rc == 0, rc

        # dense comparator: torch.matmul (hipBLASLt / rocBLAS) on the same rotating buffers
        def dense(i, bufs=bufs, x=x):
            for w in bufs[i]:
                torch.matmul(x.view(1, -1), w)
        d_med, d_lo, d_hi = time_graph(dense, nbuf, a.launches, a.reps)
        rec = {"kind": "dense_torch", "case": tag, "Z": Z, "N": N, "nmat": nmat, "us": d_med, "us_p20": d_lo, "us_p80": d_hi,
               "tbs": algo_bytes(Z, N, Z, nmat) / d_med / 1e6}
        out.write(json.dumps(rec) + "\n")
        print(f"[{tag}] Z={Z} N={N} nnz={nnz} algo={ab/1e6:.1f} MB ideal@8TB/s={ab/8e6:.2f} us | torch dense {d_med:.2f} us ({rec['tbs']:.2f} TB/s)")
        results = []
        tiles64 = (N * nmat + 63) // 64
        for lpr, wv, un in itertools.product(lprs, waves, unrolls):
            tiles = (N * nmat + lpr * 8 - 1) // (lpr * 8)
            # split candidates: total workgroups ~ 1, 2, 4, 8 x CUs
            cands = sorted({max(1, min(32, round(m * ncu / tiles))) for m in (0.5, 1, 2, 4, 8)} | {1})
            for sp in cands:
                if nnz // sp < wv * (64 // lpr):  # less than one step per workgroup: pointless
                    continue
                if L.teal_set_tuning(lpr, wv, sp, un) != 0:
                    continue
                try:
                    med, lo, hi = time_graph(launch, nbuf, a.launches, a.reps)
                except AssertionError:
                    continue
                rec = {"kind": "sparse", "case": tag, "Z": Z, "N": N, "nmat": nmat, "nnz": nnz, "s": s, "lpr": lpr, "waves": wv, "split": sp,
                       "unroll": un, "wgs": tiles * sp, "us": med, "us_p20": lo, "us_p80": hi, "tbs": ab / med / 1e6}
                out.write(json.dumps(rec) + "\n")
                results.append(rec)
        out.flush()
        results.sort(key=lambda r: r["us"])
        for r in results[:6]:
            print(f"    lpr={r['lpr']:2d} waves={r['waves']:2d} split={r['split']:2d} unroll={r['unroll']} wgs={r['wgs']:5d}  {r['us']:7.2f} us  {r['tbs']:.2f} TB/s")
        L.teal_set_tuning(0, 0, 0, 0)
        med, lo, hi = time_graph(launch, nbuf, a.launches, a.reps)
        cfg = (ctypes.c_int * 5)()
        L.teal_get_config(Z, N * nmat, 1, cfg)
        print(f"    auto {list(cfg)}: {med:.2f} us {ab / med / 1e6:.2f} TB/s   (elapsed {time.time() - t_start:.0f}s)")
        out.write(json.dumps({"kind": "auto", "case": tag, "cfg": list(cfg), "us": med, "tbs": ab / med / 1e6}) + "\n")
        del bufs, x, y
        torch.cuda.empty_cache()
    out.close()


if __name__ == "__main__":
    main()
