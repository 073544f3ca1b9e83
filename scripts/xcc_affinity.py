#!/usr/bin/env python3
"""Diagnostic: stream time of each XCC when it reads column-tile residue (xcc + rot) % 8."""
import os, sys, ctypes
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teal_amd import _lib, runtime
from _phase import legacy_view

def main():
    L = _lib.load(); runtime.init()
    dt = torch.float16
    for tag, Z, N, lpr in (("lpr8_N12288", 4096, 12288, 8), ("lpr16_N22016", 4096, 22016, 16), ("lpr32_N16384x", 4096, 16384, 32)):
        g = torch.Generator(device="cuda").manual_seed(0)
        x = (torch.rand(1, 1, Z, device="cuda", generator=g) - 0.5).to(dt)
        nbuf = int(1.2e9 / (Z * N * 2)) + 1
        bufs = [(torch.rand(Z, N, device="cuda", generator=g) - 0.5).to(dt) for _ in range(nbuf)]
        ws = runtime.reserve_workspace(Z, N); y = torch.empty(N, device="cuda", dtype=dt)
        tiles = N // (lpr * 8)
        split = max(1, 256 // tiles)
        L.teal_set_tuning(lpr, 16, split, 4)
        wgs = tiles * split
        phase = torch.zeros(wgs * 32, dtype=torch.int64, device="cuda")
        M = np.zeros((8, 8)); spans = []
        for rot in range(8):
            L.teal_set_swizzle(8 + rot)
            acc = np.zeros(8); cnt = 0; sp = []
            for it in range(6):
                phase.zero_(); torch.cuda.synchronize()
                L.teal_set_phase_buffer(phase.data_ptr())
                rc = L.teal_sparse_gemv(x.data_ptr(), bufs[it % nbuf].data_ptr(), y.data_ptr(), 0.25, Z, N, 0, ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
                assert rc == 0
                torch.cuda.synchronize(); L.teal_set_phase_buffer(None)
                if it < 2: continue
                p = legacy_view(phase, wgs).cpu().numpy()
                xcc = (p[:, 7] & 0xffffffff).astype(int)
                st = (p[:, 4] - p[:, 3]) * 0.01
                for k in range(8): acc[k] += st[xcc == k].mean()
                cnt += 1; sp.append((p[:, 5].max() - p[:, 0].min()) * 0.01)
            for k in range(8): M[k, (k + rot) % 8] = acc[k] / cnt
            spans.append(np.median(sp))
        print(f"[{tag}] tiles={tiles} split={split}; stream us, rows = XCC, cols = tile residue; spans per rot: " + " ".join(f"{v:.1f}" for v in spans))
        for k in range(8): print("   xcc%d " % k + " ".join(f"{v:6.2f}" for v in M[k]))
        L.teal_set_swizzle(0); L.teal_set_tuning(0, 0, 0, 0)
        del bufs

if __name__ == "__main__":
    main()
