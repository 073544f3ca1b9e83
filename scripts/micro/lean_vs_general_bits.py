"""Lean (teal_gemv_fast.h) vs general (teal_gemv_kernel.h) kernel, same launch, raw fp32 slabs compared bit for bit
(diagnostics build).  Run twice: weights as drawn, and with fp16 subnormal weights / activations flushed to zero — if only
the first differs, one kernel's multiply-add drops subnormal fp16 inputs and the other's does not."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from teal_amd import _lib, runtime  # noqa: E402
from teal_amd.gpt_fast.engine import TEAL_IN_PLAIN, TEAL_OUT_SLABS, GemvIn, _out  # noqa: E402

DEV = "cuda"


def launch(L, x, W, tau, code, ws):
    N, Z = W.shape
    slabs = torch.zeros(8, N, dtype=torch.float32, device=DEV)
    gin = GemvIn(mode=TEAL_IN_PLAIN, x=x.data_ptr())
    gout = _out([(W.data_ptr(), W.stride(1), 0, N, float(tau), None)], TEAL_OUT_SLABS, slabs)
    dbuf = ctypes.create_string_buffer(160)
    gout.desc, gout.desc_bytes = ctypes.cast(dbuf, ctypes.c_char_p), 160
    n = ctypes.c_int(0)
    _lib.check(L.teal_fused_gemv(ctypes.byref(gin), ctypes.byref(gout), Z, code, ws.data_ptr(), ws.numel() * 4, ctypes.byref(n),
                                 runtime.stream_ptr()), "teal_fused_gemv")
    torch.cuda.synchronize()
    st = (n.value + 3) & ~3
    return slabs.view(-1)[: N * st].view(N, st)[:, : n.value].clone(), dbuf.value.decode()


def main():
    with _lib.diagnostics() as L:
        ws = torch.zeros(1 << 20, dtype=torch.int32, device=DEV)
        L.teal_workspace_init(ws.data_ptr(), ws.numel() * 4, runtime.stream_ptr())
        for code, dt in ((0, torch.float16), (1, torch.bfloat16)):
            for Z, N in ((4096, 12288), (4096, 4096), (11008, 4096)):
                for flush in (False, True):
                    g = torch.Generator(device=DEV).manual_seed(Z + N)
                    x = (torch.randn(Z, device=DEV, generator=g) * 1.0).to(dt)
                    buf = (torch.randn(Z, N + 64, device=DEV, generator=g) * 0.02).to(dt)
                    tiny = torch.finfo(dt).tiny
                    nsub_w = int((buf.float().abs() < tiny).sum() - (buf == 0).sum())
                    nsub_x = int((x.float().abs() < tiny).sum() - (x == 0).sum())
                    if flush:
                        buf = torch.where(buf.float().abs() < tiny, torch.zeros_like(buf), buf)
                        x = torch.where(x.float().abs() < tiny, torch.zeros_like(x), x)
                    W = buf[:, :N].T
                    tau = float(x.float().abs().median())
                    outs = {}
                    for fast in (1, 0):
                        L.teal_set_fast(fast)
                        outs[fast] = launch(L, x, W, tau, code, ws)
                    L.teal_set_fast(1)
                    (a, da), (b, db) = outs[1], outs[0]
                    same_shape = a.shape == b.shape
                    ndiff = int((a.view(torch.int32) != b.view(torch.int32)).sum()) if same_shape else -1
                    sa, sb = a.sum(1), b.sum(1)
                    print(f"{dt} {Z}x{N} flush={flush} subnormal w {nsub_w} x {nsub_x}: slabs lean {tuple(a.shape)} general {tuple(b.shape)} "
                          f"differing words {ndiff}; max |sum diff| {float((sa - sb).abs().max()):.3e}\n    lean: {da}\n    general: {db}")


main()
