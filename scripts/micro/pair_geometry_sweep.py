"""Paired gate | up launch (teal_sparse_gateup_silu: both matrices in one workgroup per column tile) at 50 %: the automatic tile
width against forced 64 / 128 / 256-column tiles (diagnostics build), for widths whose tile count falls between one and two
rounds of workgroups (Llama-30B: inter 17920 = 280 x 64 columns on 256 CUs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from teal_amd import _lib  # noqa: E402
from teal_amd import kernels as K  # noqa: E402

DEV = "cuda"
shapes = [("7B", 4096, 11008), ("13B", 5120, 13824), ("30B", 6656, 17920), ("34B", 8192, 22016), ("70B", 8192, 28672)]
with _lib.diagnostics() as L:
    for name, Z, N in shapes:
        g = torch.Generator(device=DEV).manual_seed(Z)
        x = torch.randn(1, 1, Z, device=DEV, dtype=torch.float16, generator=g)
        tau = float(x.abs().median())
        sets = []
        for _ in range(2):  # two weight sets, alternated: 2 x 2 x Z x N x 2 bytes > the 256 MB Infinity Cache from 13B on
            sets.append(tuple((torch.randn(Z, N, device=DEV, dtype=torch.float16, generator=g) * 0.02).T for _ in range(2)))
        line = [f"{name:4s} Z {Z:5d} N {N:5d} algo {2 * 0.5 * Z * N * 2 / 1e6:7.1f} MB |"]
        for lpr in (0, 8, 16, 32):
            assert L.teal_set_tuning(lpr, 0, 0, 0) == 0
            for w1, w3 in sets:
                K.sparse_gateup_silu(x, w1, w3, tau, tau)
            torch.cuda.synchronize()
            n = 40
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                w1, w3 = sets[i & 1]
                K.sparse_gateup_silu(x, w1, w3, tau, tau)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            desc = L.teal_last_launch_desc().decode()
            grid = desc.split("grid")[1].strip() if "grid" in desc else desc
            line.append(f" lpr {lpr or 'auto':>4}: {us:6.1f} us {2 * 0.5 * Z * N * 2 / us / 1e6:5.2f} TB/s [{grid}] |")
        L.teal_set_tuning(0, 0, 0, 0)
        print("".join(line))
        del sets
        torch.cuda.empty_cache()
