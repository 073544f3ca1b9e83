#!/usr/bin/env python3
"""Where an int4 decode launch spends its time: phase stamps of sparse_gemv_int4_kernel<.., PHASE = true>
(teal_gemv_int4.hip) for the four GEMV launches of a decode layer at Llama-2-7B widths, int4-g32 weights, 50 % activation
sparsity.  One eager engine step over `--layers` distinct layers (so the weights come from HBM, not the Infinity Cache)
with teal_set_phase_buffer / teal_set_phase_stride; per launch role the median over workgroups and layers of every stamp
relative to the workgroup's entry, and the launch's extent (first entry -> last end over its workgroups).

Stamps (thread 0, 100 MHz wall clock -> 10 ns): 1 arguments in registers, 2 producer done, 3 activations of the first pass
ready, 4 list written, 5 every load of the pass issued, 6 first unit consumed, 7 pass done, 8 all passes done, 9 past the
reduce barrier, 10 outputs stored."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TEAL_LIB_FLAVOR", "diag")  # the tuning / phase-stamp switches exist in libteal_hip_diag.so only
from teal_amd import _lib, runtime  # noqa: E402
from teal_amd.gpt_fast import generate as G  # noqa: E402
from teal_amd.gpt_fast.engine import DecodeEngine  # noqa: E402
from teal_amd.quantize import quantize_model_int4  # noqa: E402

ROW = 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7B")
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--sparsity", type=float, default=0.5)
    ap.add_argument("--groupsize", type=int, default=32)
    a = ap.parse_args()
    L = _lib.load()
    runtime.init()
    dev = "cuda"
    model = quantize_model_int4(G.build_synthetic_model(a.model, dev, torch.float16, n_layer=a.layers), a.groupsize)
    ths = G.apply_sparsity(model, sparsity=a.sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
    model.max_seq_length = -1
    model.setup_caches(1, 256)
    prompt = torch.randint(0, model.config.vocab_size, (120,), device=dev, dtype=torch.int)
    with torch.no_grad():
        model(prompt.view(1, -1), torch.arange(120, device=dev))
    eng = DecodeEngine(model, ths)
    tok = torch.tensor([[7]], device=dev, dtype=torch.int)
    pos = torch.tensor([120], device=dev, dtype=torch.int)
    for _ in range(3):
        eng(tok, pos)
    torch.cuda.synchronize()
    stride = ROW * 512
    nlaunch = 5 * a.layers + 1
    buf = torch.zeros(stride * (nlaunch + 2), device=dev, dtype=torch.int64)
    reps = []
    for _ in range(5):
        buf.zero_()
        L.teal_set_phase_buffer(buf.data_ptr())
        L.teal_set_phase_stride(stride)
        eng(tok, pos)
        torch.cuda.synchronize()
        L.teal_set_phase_buffer(None)
        L.teal_set_phase_stride(0)
        reps.append(buf.clone().cpu())
    roles = {"qkv (MODE 1)": 0, "wo (MODE 4)": 2, "gate|up (MODE 1)": 3, "down (MODE 2)": 4}
    print(f"{a.model} widths, int4-g{a.groupsize}, sparsity {a.sparsity}, {a.layers} layers, 5 eager steps; us after the workgroup's entry (median over workgroups x layers x steps)")
    print("%-18s %5s %7s | %s" % ("launch", "wgs", "extent", "  ".join(f"[{i:2d}]" for i in range(1, 11))))
    for name, off in roles.items():
        rel, ext, wgs = [], [], 0
        for b in reps:
            for layer in range(1, a.layers):
                r = b[(layer * 5 + off) * stride: (layer * 5 + off + 1) * stride].view(512, ROW)
                live = r[:, 0] > 0
                wgs = int(live.sum())
                r = r[live]
                rel.append((r[:, 1:11] - r[:, :1]).float() / 100.0)
                ext.append(float(r[:, 10].max() - r[:, 0].min()) / 100.0)
        rel = torch.cat(rel)
        med = rel.median(dim=0).values
        print("%-18s %5d %7.2f | %s" % (name, wgs, sorted(ext)[len(ext) // 2], "  ".join("%4.1f" % float(v) for v in med)))


if __name__ == "__main__":
    main()
