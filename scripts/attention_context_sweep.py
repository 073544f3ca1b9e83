#!/usr/bin/env python3
"""Decode attention (split-KV + merge launch) against context length, split count and head grouping.

Per call: K/V bytes the algorithm needs once (n_kv x positions x head_dim x 2 tensors x 2 B) over the measured time, i.e.
the fraction of the HBM roofline the kernel reaches when the cache no longer fits the last-level cache (the caches rotate
over > 512 MB).  With grouped-query attention every K/V row is used by n_head / n_kv query heads.
"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teal_amd import _lib, runtime
from teal_amd.gpt_fast.model import precompute_freqs_cis


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splits", type=int, nargs="*", default=[4, 8, 16, 32])
    ap.add_argument("--voff", type=int, default=0, help="shift every V cache this many bytes into its allocation (DRAM channel phase of K vs V)")
    ap.add_argument("--models", nargs="*", default=["7B", "8B", "70B"])
    ap.add_argument("--ctx", type=int, nargs="*", default=[1024, 4096, 16384])
    ap.add_argument("--pos", type=int, default=0, help="decode position inside the cache (default: the cache's last but one row); a position well "
                    "inside a long cache shows what a launch reads past the sequence")
    ap.add_argument("--fold", type=int, default=0, help="1: hand the launch a prepared workspace (the merge folded in by arrival tickets)")
    a = ap.parse_args()
    L = _lib.load(); runtime.init()
    global WSP
    WSP = runtime.new_workspace(64, 64)
    dt = torch.float16
    hd = 128
    for name, n_head, n_kv in (("7B", 32, 32), ("8B", 32, 8), ("70B", 64, 8)):
        if name not in a.models:
            continue
        for S in a.ctx:
            pos = min(a.pos, S - 2) if a.pos > 0 else S - 2
            nrot = max(2, min(16, int(600e6 // (n_kv * S * hd * 4)) + 1))
            kcs = [torch.randn(n_kv, S, hd, device="cuda").to(dt) for _ in range(nrot)]
            vcs = []
            for _ in range(nrot):
                flat = torch.empty(n_kv * S * hd + a.voff // 2, device="cuda", dtype=dt)
                v = flat[a.voff // 2:].view(n_kv, S, hd)
                v.copy_(torch.randn(n_kv, S, hd, device="cuda").to(dt))
                vcs.append(v)
            qkv = torch.randn((n_head + 2 * n_kv) * hd, device="cuda").to(dt)
            rope = precompute_freqs_cis(S, hd, 10000, dt).cuda().contiguous()
            y = torch.empty(n_head * hd, device="cuda", dtype=dt)
            p = torch.tensor([pos], device="cuda", dtype=torch.int32)
            kv_bytes = n_kv * (pos + 1) * hd * 2 * 2
            ptag = f" pos {pos}" if a.pos > 0 else ""
            line = f"{name:>3} heads {n_head}/{n_kv} ctx {S:>5}{ptag} ({kv_bytes / 1e6:6.1f} MB K/V)"
            for ns in a.splits:
                ws = torch.zeros(n_head * ns * (hd + 2), device="cuda", dtype=torch.float32)
                st = torch.cuda.Stream()
                def call(i):
                    return L.teal_decode_attention_split_ws(qkv.data_ptr(), None, 0, rope.data_ptr(), p.data_ptr(), kcs[i % nrot].data_ptr(),
                                                            vcs[i % nrot].data_ptr(), y.data_ptr(), None, 0.0, n_head, n_kv, hd, S, ns,
                                                            ws.data_ptr(), ws.numel() * 4, 0, None,
                                                            WSP.numel() * 4 if a.fold else 0, st.cuda_stream)
                with torch.cuda.stream(st):
                    rc = call(0)
                    if rc != 0:
                        line += f" | x{ns}: rc {rc}"
                        continue
                    st.synchronize()
                    # back-to-back launches inside one hipGraph, as in the decode engine (eager launches add ~6 us per kernel)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st):
                        for i in range(nrot): call(i)
                    g.replay(); st.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    reps = 4
                    e0.record(st)
                    for _ in range(reps): g.replay()
                    e1.record(st)
                    st.synchronize()
                    n = reps * nrot
                us = e0.elapsed_time(e1) * 1e3 / n
                line += f" | x{ns}: {us:7.1f} us {kv_bytes / us / 1e6:5.2f} TB/s"
            print(line, flush=True)


if __name__ == "__main__":
    main()
