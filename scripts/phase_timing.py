#!/usr/bin/env python3
"""Where does a sparse-GEMV launch spend its time?  Uses teal_set_phase_buffer(): thread 0 of every
workgroup stamps a 100 MHz wall clock at phase boundaries.  Benchmark utility (GPU box)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TEAL_LIB_FLAVOR", "diag")  # the tuning / phase-stamp switches exist in libteal_hip_diag.so only
from teal_amd import _lib, runtime
from _phase import legacy_view  # noqa: E402

NAMES = ["start->ballots", "ballots->scatter", "scatter->barrier", "rows streamed", "reduce+store"]


def main():
    L = _lib.load()
    runtime.init()
    dt = torch.float16
    cases = [("wo_7b", 4096, 4096, 0.5, 1), ("gate_7b", 4096, 11008, 0.5, 1), ("gateup_7b", 4096, 11008, 0.5, 2),
             ("down_7b", 11008, 4096, 0.5, 1), ("lmhead", 4096, 32000, 0.0, 1), ("wo_dense", 4096, 4096, 0.0, 1)]
    int8 = os.environ.get("TEAL_WEIGHTS", "") == "int8"  # int8 weight-only variant of the single-matrix cases
    if int8:
        cases = [c for c in cases if c[4] == 1]
    for tag, Z, N, s, nmat in cases:
        g = torch.Generator(device="cuda").manual_seed(0)
        x = (torch.rand(1, 1, Z, device="cuda", generator=g) - 0.5).to(dt)
        tau = s / 2 if s > 0 else -1.0
        nbuf = max(2, int(1.1e9 / (Z * N * 2 * nmat)) + 1)
        if int8:
            nbuf = max(2, int(1.1e9 / (Z * N)) + 1)
            bufs = [[torch.randint(-127, 128, (Z, N), device="cuda", generator=g, dtype=torch.int8)] for _ in range(nbuf)]
            sc = torch.full((N,), 1e-3, device="cuda", dtype=dt)
        else:
            bufs = [[(torch.rand(Z, N, device="cuda", generator=g) - 0.5).to(dt) for _ in range(nmat)] for _ in range(nbuf)]
        ws = runtime.reserve_workspace(Z, N)
        y = torch.empty(N * nmat, device="cuda", dtype=dt)
        cfg = (ctypes.c_int * 5)()
        L.teal_get_config(Z, N * nmat, 1, cfg)
        wgs = cfg[4]
        phase = torch.zeros(wgs * 32, dtype=torch.int64, device="cuda")
        spans, rows = [], []
        for it in range(12):
            phase.zero_()
            torch.cuda.synchronize()
            L.teal_set_phase_buffer(phase.data_ptr())
            b = bufs[it % nbuf]
            if int8:
                rc = L.teal_sparse_qkv_gemv_i8(x.data_ptr(), b[0].data_ptr(), sc.data_ptr(), y.data_ptr(), tau, tau, tau, Z, N, N, 0, N, 0,
                                               ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
            elif nmat == 1:
                rc = L.teal_sparse_gemv(x.data_ptr(), b[0].data_ptr(), y.data_ptr(), tau, Z, N, 0, ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
            else:
                rc = L.teal_sparse_gateup_silu(x.data_ptr(), b[0].data_ptr(), b[1].data_ptr(), y.data_ptr(), tau, tau, Z, N, 0, ws.data_ptr(),
                                               ws.numel() * 4, runtime.stream_ptr())
            assert rc == 0
            torch.cuda.synchronize()
            L.teal_set_phase_buffer(None)
            if it < 2:
                continue
            p = legacy_view(phase, wgs).cpu().double() * 10.0  # ns
            t0 = p[:, 0].min()
            spans.append(float(p[:, 5].max() - t0) / 1e3)
            rows.append([float((p[:, 0] - t0).max()) / 1e3] + [float((p[:, i + 1] - p[:, i]).mean()) / 1e3 for i in range(5)] +
                        [float((p[:, 5] - t0).min()) / 1e3, float((p[:, 6] - p[:, 1]).mean()) / 1e3, 0.0,
                         float((p[:, 2] - p[:, 6]).mean()) / 1e3])
        import numpy as np
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.save(os.path.join(ROOT, "gpurun_out", f"phase_{tag}.npy"), (legacy_view(phase, wgs).cpu().numpy() - int(legacy_view(phase, wgs)[:, 0].min())))
        r = torch.tensor(rows).median(dim=0).values.tolist()
        print(f"[{tag}] cfg={list(cfg)} span(first start -> last end) median {sorted(spans)[len(spans) // 2]:.2f} us")
        print(f"    dispatch skew (last WG start) {r[0]:.2f} us; earliest WG end {r[6]:.2f} us")
        print(f"    [ballots->scatter = barrier wait {r[7]:.2f} + popcount total {r[8]:.2f} + scan/scatter {r[9]:.2f}]")
        for n, v in zip(NAMES, r[1:6]):
            print(f"    {n:18s} {v:6.2f} us (mean over WGs)")
        del bufs


if __name__ == "__main__":
    main()
