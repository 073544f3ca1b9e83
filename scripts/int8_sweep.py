#!/usr/bin/env python3
"""int8 vs 16-bit sparse GEMV, per launch geometry (GPU box).  Weights rotate over > 1 GB of buffers inside a
hipGraph of back-to-back launches (kernel + one same-stream boundary).  Benchmark utility."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TEAL_LIB_FLAVOR", "diag")  # the tuning / phase-stamp switches exist in libteal_hip_diag.so only
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from teal_amd import _lib, runtime  # noqa: E402
from teal_amd.kernels import sparse_gemv as K  # noqa: E402
from tune_gemv import time_graph  # noqa: E402


def main():
    L = _lib.load()
    runtime.init()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else \
        ((4096, 11008), (4096, 12288), (11008, 4096), (4096, 4096), (8192, 28672))
    for Z, N in shapes:
        x = (torch.rand(1, 1, Z, device=dev, generator=g) - 0.5).half()
        tau = 0.25
        nnz = int((x.float().abs() > tau).sum())
        for wbytes, pad in ((2, 64), (1, 128)):
            nbuf = max(2, int(1.1e9 / (Z * N * wbytes)) + 1)
            bufs = []
            for _ in range(nbuf):
                b = torch.zeros(Z, N + pad, device=dev, dtype=torch.float16 if wbytes == 2 else torch.int8)
                if wbytes == 2:
                    b[:, :N] = (torch.rand(Z, N, device=dev, generator=g) - 0.5).half()
                else:
                    b[:, :N] = torch.randint(-127, 128, (Z, N), device=dev, generator=g, dtype=torch.int8)
                bufs.append(b[:, :N].T)
            sc = torch.full((N,), 1e-3, device=dev, dtype=torch.float16)
            algo = nnz * N * wbytes + Z * 2 + N * 2
            for lpr in (0, 8, 16, 32):
                for split in (0, 1, 2, 4, 8):
                    if L.teal_set_tuning(lpr, 0, split, 0) != 0:
                        continue

                    def launch(i):
                        if wbytes == 2:
                            K.splitk_sparse_gemv(x, bufs[i], tau, 0)
                        else:
                            K.splitk_sparse_gemv_int8(x, bufs[i], sc, tau, 0)
                    try:
                        med, lo, hi = time_graph(launch, nbuf, 48, 7)
                    except Exception as e:  # geometry not available for this shape
                        print(f"{Z}x{N} w{wbytes} lpr={lpr} split={split}: {type(e).__name__}")
                        continue
                    print(f"{Z}x{N} w{wbytes}B lpr={lpr:2d} split={split}: {med:6.2f} us  ({algo / med / 1e6:5.2f} TB/s algorithmic)", flush=True)
            L.teal_set_tuning(0, 0, 0, 0)
            del bufs
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
