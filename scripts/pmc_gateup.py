#!/usr/bin/env python3
"""The dominant launch of the decode step — fused RMSNorm -> mask -> gate | up GEMV — on its own, for a rocprofv3 counter pass.

bench.py runs this file under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (a subprocess of the bench run itself: PMC
counters cannot be sampled from inside a process) and reads the HBM bytes the launch fetched, so that `roofline.traffic`
is measured in the same run as `roofline.achieved`.  Same entry point (teal_fused_gemv), same geometry, same kept fraction
as the engine's launch: K weight pairs of the model's shape (rotated: K x 180 MB >> the 256 MB Infinity Cache), a residual
+ 4 interleaved slabs + norm weight as the producer's input, thresholds at the `sparsity` quantile of |x| (kept fraction
1 - sparsity: 0.6 for Llama-3-8B @ 40 %, not the median).
Prints one JSON line: kernel instantiation, grid, kept fraction, algorithmic bytes per launch."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teal_amd import _lib, runtime  # noqa: E402
from teal_amd.gpt_fast.engine import GemvIn, GemvOut, TEAL_IN_RESID_NORM, TEAL_OUT_PAIR_SILU, TEAL_OUT_ROUNDED  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--inter", type=int, default=11008)
    ap.add_argument("--sparsity", type=float, default=0.5)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--sets", type=int, default=4)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--pair", type=int, default=0, help="1: the paired gate|up launch (silu * up + keep masks in its epilogue), as the engine runs 70B-class widths")
    a = ap.parse_args()
    L = _lib.load()
    runtime.init()
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    code = runtime.dtype_code(dt)
    Z, N, ld = a.dim, a.inter, a.inter + 64
    g = torch.Generator(device="cuda").manual_seed(3)
    mk = lambda: ((torch.rand(Z, ld, device="cuda", generator=g) - 0.5) * 0.04).to(dt)  # noqa: E731
    ws_ = [(mk(), mk()) for _ in range(a.sets)]
    resid = (torch.randn(Z, device="cuda", generator=g)).to(dt)
    slabs = (torch.randn(Z, 4, device="cuda", generator=g) * 0.25).float().contiguous()
    normw = torch.ones(Z, device="cuda", dtype=dt)
    # the producer's x, restated with torch, for the threshold (kept fraction 1 - sparsity)
    y = slabs[:, 0]
    for j in range(1, 4):
        y = y + slabs[:, j]
    h = (resid.float() + y.to(dt).float()).to(dt).float()
    x = ((h * torch.rsqrt(h.pow(2).mean() + 1e-5)).to(dt) * normw).float().abs()
    tau = float(torch.quantile(x, a.sparsity)) if a.sparsity > 0 else -1.0  # kept fraction 1 - sparsity (the median only at 0.5)
    kept = float((x > tau).float().mean())
    gu = torch.empty(2 * N, device="cuda", dtype=dt)
    hmask = torch.zeros((N + 63) // 64, device="cuda", dtype=torch.int64)
    wsb = runtime.new_workspace(Z, 2 * N)
    gin = GemvIn(mode=TEAL_IN_RESID_NORM, resid_in=resid.data_ptr(), slabs=slabs.data_ptr(), nslabs=4, slabs_interleaved=1,
                 norm_weight=normw.data_ptr(), eps=1e-5, resid_out=None)
    desc = None
    dbuf = ctypes.create_string_buffer(160)  # the launch reports its own kernel instantiation + grid (teal_gemv_out_t.desc)
    for _ in range(a.reps):
        for w1, w3 in ws_:
            o = GemvOut()
            o.nseg, o.mode, o.act_seg0 = 2, (TEAL_OUT_PAIR_SILU if a.pair else TEAL_OUT_ROUNDED), (0 if a.pair else 1)
            if a.pair:
                o.mask_out, o.mask_tau = hmask.data_ptr(), 0.01
            for i, w in enumerate((w1, w3)):
                o.w[i], o.ld[i], o.col0[i], o.ncols[i], o.tau[i] = w.data_ptr(), ld, 0, N, tau
                o.y[i] = gu.data_ptr() + 2 * N * i
            o.desc, o.desc_bytes = ctypes.cast(dbuf, ctypes.c_char_p), 160
            rc = L.teal_fused_gemv(ctypes.byref(gin), ctypes.byref(o), Z, code, wsb.data_ptr(), wsb.numel() * 4, None, runtime.stream_ptr())
            assert rc == 0, rc
            desc = dbuf.value.decode()
    torch.cuda.synchronize()
    nnz = int((x > tau).sum())
    print(json.dumps({"kernel": desc, "launches": a.sets * a.reps, "kept_fraction": kept,
                      "algorithmic_bytes": 2 * nnz * N * 2 + Z * 2 + (N if a.pair else 2 * N) * 2,  # kept rows + x + y
                      "producer_bytes": Z * 2 + 4 * Z * 4 + Z * 2 + ((N // 8) if a.pair else 0)}))


if __name__ == "__main__":
    main()
