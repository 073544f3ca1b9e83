#!/usr/bin/env python3
"""BASELINE config 1 (single fp16 GEMV 4096 -> 4096 at 50 % activation sparsity): per-call time of every launch geometry the
library can be forced into (teal_set_tuning: lanes per row segment x split-K factor), same timing as scripts/benchmark_gemv.py
(hipGraph of 32 back-to-back launches rotating over > 1 GB of distinct weights; per call incl. the launch boundary).
Answers the round-3 verdict's item 2: is there a geometry without the arrival-ticket tail that gets this shape under 6 us?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TEAL_LIB_FLAVOR", "diag")  # the tuning / phase-stamp switches exist in libteal_hip_diag.so only
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from teal_amd import _lib, runtime  # noqa: E402
from benchmark_gemv import graph_times  # noqa: E402


def main():
    L = _lib.load()
    runtime.init()
    Z = N = 4096
    dt = torch.float16
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.rand(1, 1, Z, device="cuda", generator=g) - 0.5).to(dt)
    ld = N + 64
    nbuf = int(1.1e9 / (Z * ld * 2)) + 1
    bufs = []
    for _ in range(nbuf):
        b = torch.zeros(Z, ld, device="cuda", dtype=dt)
        b[:, :N] = (torch.rand(Z, N, device="cuda", generator=g) - 0.5).to(dt)
        bufs.append(b)
    ws = runtime.reserve_workspace(Z, N)
    y = torch.empty(N, device="cuda", dtype=dt)
    for s_ in (0.5, 0.99):
        tau = s_ / 2
        nnz = int((x.float().abs() > tau).sum())
        print(f"-- sparsity {s_:.2f} (nnz {nnz}): {nnz * N * 2 / 1e6:.1f} MB of kept rows; HBM time at 6.8 TB/s {nnz * N * 2 / 6.8e12 * 1e6:.2f} us")

        def sparse(i):
            rc = L.teal_sparse_qkv_gemv_ld(x.data_ptr(), bufs[i % nbuf].data_ptr(), ld, y.data_ptr(), tau, tau, tau, Z, N, N, 0, 0,
                                           ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
            assert rc == 0, rc

        for lpr, split in ((0, 0), (8, 1), (8, 2), (8, 4), (8, 8), (16, 1), (16, 2), (16, 4), (16, 8), (32, 8), (64, 8), (64, 16)):
            assert L.teal_set_tuning(lpr, 0, split, 0) == 0
            ms = graph_times(sparse, 32, 9)
            desc = L.teal_last_launch_desc().decode()
            print(f"   lanes {lpr:2d} split {split:2d}: {ms[0] * 1e3:6.2f} us per call (p20 {ms[1] * 1e3:.2f}, p80 {ms[2] * 1e3:.2f})   {desc}")
        L.teal_set_tuning(0, 0, 0, 0)


if __name__ == "__main__":
    main()
