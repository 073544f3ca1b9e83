#!/usr/bin/env python3
"""Phase timing of the fused engine launches (RESID_NORM / SILU_MUL producers) on 7B shapes.
Benchmark utility (GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teal_amd import _lib, runtime  # noqa: E402
from teal_amd.gpt_fast import generate as G  # noqa: E402
from teal_amd.gpt_fast.engine import DecodeEngine  # noqa: E402


def main():
    L = _lib.load()
    runtime.init()
    L.teal_set_swizzle(int(os.environ.get("TEAL_SWIZZLE", "0")))
    model = G.build_synthetic_model("7B", "cuda", torch.float16, n_layer=6)
    if os.environ.get("TEAL_WEIGHTS", "") == "int8":
        from teal_amd.quantize import quantize_model_int8
        quantize_model_int8(model)
    ths = G.apply_sparsity(model, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True)
    model.max_seq_length = -1
    model.setup_caches(1, 64)
    prompt = torch.randint(0, 32000, (6,), device="cuda", dtype=torch.int)
    with torch.no_grad():
        model(prompt.view(1, -1), torch.arange(6, device="cuda"))
    eng = DecodeEngine(model, ths, pair=(None if "TEAL_PAIR" not in os.environ else bool(int(os.environ["TEAL_PAIR"]))))
    tok = torch.tensor([[5]], device="cuda", dtype=torch.int)
    pos = torch.tensor([6], device="cuda", dtype=torch.int)
    for _ in range(3):
        eng(tok, pos)
    torch.cuda.synchronize()
    phase = torch.zeros(2048 * 24, dtype=torch.int64, device="cuda")
    k1_in, k1_out, kc, vc, k3_in, k3_out, k4_in, k4_out, k5_in, k5_out, _tau_o = eng.stages[3]
    k1_in.nslabs = eng.n_down.value
    k4_in.nslabs = eng.n_wo.value
    print("slabs: wo", eng.n_wo.value, "down", eng.n_down.value)
    names = ["load+norm+ballots (0->1)", "barrier wait (1->6)", "scan/scatter (6->2)", "-", "barrier (2->3)", "rows streamed (3->4)",
             "reduce+store (4->5)"]
    order = [(0, 1), (1, 6), (6, 2), (2, 2), (2, 3), (3, 4), (4, 5)]
    for tag, gin, gout, Z in (("qkv  [RESID_NORM]", k1_in, k1_out, eng.dim), ("wo   [MASKED]", k3_in, k3_out, eng.dim),
                              ("g|u  [PAIR]", k4_in, k4_out, eng.dim), ("down [MASKED]", k5_in, k5_out, eng.inter)):
        spans, rows = [], []
        for it in range(8):
            # evict the weights from the Infinity Cache between repeats: run the other layers
            eng(tok, pos)
            phase.zero_()
            torch.cuda.synchronize()
            L.teal_set_phase_buffer(phase.data_ptr())
            eng._stream = runtime.stream_ptr()
            eng._gemv(gin, gout, Z)
            torch.cuda.synchronize()
            L.teal_set_phase_buffer(None)
            p = phase[: 2048 * 8].view(-1, 8)
            n = int((p[:, 5] > 0).sum()) // 3  # n rows of phase stamps + 2n rows' worth of per-wave stamps
            p = p[:n]
            pw = phase[n * 8: n * 8 + n * 16].view(n, 16).cpu().double() * 10.0 / 1e3  # per-wave end of stream, us
            p3 = p[:n, 3].cpu().double() * 10.0 / 1e3
            if it == 7:
                print('      per-wave mean stream us:', ' '.join(f'{v:.1f}' for v in (pw - p3[:, None]).mean(dim=0).tolist()))
            wave_spread = float((pw.max(dim=1).values - pw.min(dim=1).values).mean())
            wave_mean_dur = float((pw.mean(dim=1) - p3).mean())
            wave_max_dur = float((pw.max(dim=1).values - p3).mean())
            if it == 7:
                import numpy as np
                np.save(os.path.join(ROOT, "gpurun_out", f"ephase_{tag.split()[0].replace('|', '')}.npy"), p[:n].cpu().numpy())
            p = p[:n].cpu().double() * 10.0
            t0 = p[:, 0].min()
            spans.append(float(p[:, 5].max() - t0) / 1e3)
            rows.append([float((p[:, 0] - t0).max()) / 1e3] + [float((p[:, b] - p[:, a]).mean()) / 1e3 for a, b in order] +
                        [float((p[:, 5] - t0).min()) / 1e3])
        r = torch.tensor(rows).median(dim=0).values.tolist()
        print(f"    per-wave stream time inside a WG: mean {wave_mean_dur:.2f} us, slowest wave {wave_max_dur:.2f} us, spread {wave_spread:.2f} us")
        print(f"[{tag}] wgs={n} span {sorted(spans)[len(spans) // 2]:.2f} us; dispatch skew {r[0]:.2f}; earliest end {r[-1]:.2f}")
        for nm, v in zip(names, r[1:8]):
            print(f"    {nm:28s} {v:6.2f} us")


if __name__ == "__main__":
    main()
