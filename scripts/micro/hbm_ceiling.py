#!/usr/bin/env python3
"""Measured HBM ceilings of this MI355X next to the nominal 8 TB/s: device-to-device copy (read + write)
and a long dense GEMV through this library's own kernel (read-only stream of a 1 GiB matrix)."""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from teal_amd import _lib, runtime

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

n = 2 << 30  # 2 GiB per buffer
a = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
b = torch.empty_like(a)
t = timed(lambda: b.copy_(a))
print(f"device-to-device copy   2 GiB: {t * 1e3:7.3f} ms  read+write {2 * n / t / 1e12:5.2f} TB/s")
L = _lib.load(); runtime.init()
Z, N = 4096, 131072  # W^T [Z][N] fp16 = 1 GiB, every row kept (the shape of a 128 k-entry lm_head)
w = a.view(torch.float16)[: Z * N]
x = (torch.randn(Z, device="cuda") * 0.1).to(torch.float16)
y = torch.empty(N, device="cuda", dtype=torch.float16)
ws = runtime.reserve_workspace(Z, N)
st = runtime.stream_ptr()
def gemv():
    rc = L.teal_dense_gemv(x.data_ptr(), w.data_ptr(), y.data_ptr(), Z, N, 0, ws.data_ptr(), ws.numel() * 4, st)
    assert rc == 0, rc
t = timed(gemv)
print(f"teal_dense_gemv {Z}x{N} fp16 (1 GiB): {t * 1e3:7.3f} ms  read       {Z * N * 2 / t / 1e12:5.2f} TB/s")
