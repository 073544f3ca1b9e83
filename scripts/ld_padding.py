#!/usr/bin/env python3
"""Diagnostic: does a padded row stride (ld = N + pad) that rotates a tile over the 128-B channel
residues beat ld = N?  Per-WG stream time spread + graph time per launch."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TEAL_LIB_FLAVOR", "diag")  # the tuning / phase-stamp switches exist in libteal_hip_diag.so only
from teal_amd import _lib, runtime
from _phase import legacy_view

def graph_time(fn, n, reps=9):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn(0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(ts))

def main():
    L = _lib.load(); runtime.init()
    dt = torch.float16
    for tag, Z, N in (("qkv_12288", 4096, 12288), ("gate_11008", 4096, 11008), ("wo_4096", 4096, 4096), ("down_4096", 11008, 4096), ("lm_32000", 4096, 32000)):
        g = torch.Generator(device="cuda").manual_seed(0)
        x = (torch.rand(1, 1, Z, device="cuda", generator=g) - 0.5).to(dt)
        tau = 0.25 if "lm" not in tag else -1.0
        for pad in (0, 64, 192, 320):
            ld = N + pad
            nbuf = int(1.2e9 / (Z * ld * 2)) + 1
            bufs = [(torch.rand(Z, ld, device="cuda", generator=g) - 0.5).to(dt) for _ in range(nbuf)]
            ws = runtime.reserve_workspace(Z, N); y = torch.empty(N, device="cuda", dtype=dt)
            def launch(i):
                rc = L.teal_sparse_qkv_gemv_ld(x.data_ptr(), bufs[i % nbuf].data_ptr(), ld, y.data_ptr(), tau, tau, tau, Z, N, N, 0, 0,
                                               ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
                assert rc == 0
            t = graph_time(launch, 32)
            import ctypes
            cfg = (ctypes.c_int * 5)(); L.teal_get_config(Z, N, 1, cfg)
            wgs = cfg[4]
            phase = torch.zeros(wgs * 32, dtype=torch.int64, device="cuda")
            sp, smin, smax = [], [], []
            for it in range(6):
                phase.zero_(); torch.cuda.synchronize(); L.teal_set_phase_buffer(phase.data_ptr())
                launch(it); torch.cuda.synchronize(); L.teal_set_phase_buffer(None)
                p = legacy_view(phase, wgs).cpu().numpy()
                st = (p[:, 4] - p[:, 3]) * 0.01
                sp.append((p[:, 5].max() - p[:, 0].min()) * 0.01); smin.append(st.min()); smax.append(st.max())
            print(f"[{tag}] ld=N+{pad:3d}: graph {t:6.2f} us/launch  span {np.median(sp):6.2f}  stream min {np.median(smin):5.2f} max {np.median(smax):5.2f}")
            del bufs

if __name__ == "__main__":
    main()
