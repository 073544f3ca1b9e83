#!/usr/bin/env python3
"""What exactly changes in a sparse GEMV launch that runs next to another process's skinny GEMM?  (round 6)
One decode step of a 2-layer Llama-2-7B-width model alone on the GPU (reference), then the aggressor process starts
(concurrency_determinism_probe.py --noise-child op_linear_qkv) and ONLY layer 0's down projection is relaunched on the unchanged
inputs.  Every differing output group (slab, eight/sixteen columns at stride 8) is explained as a linear combination of weight rows:
delta[cols] ~ a * W[r, cols] for the single best row r — a dropped, doubled or foreign term shows as one row with a = -x_r, +x_r, ..."""
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine
    dev = "cuda"
    model = G.build_synthetic_model("7B", dev, torch.float16, seed=11, n_layer=2)
    ths = G.apply_sparsity(model, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True, decode_calibration=False)
    P = 6
    prompt = torch.randint(0, 32000, (P,), device=dev, dtype=torch.int, generator=torch.Generator(device=dev).manual_seed(2))
    model.max_seq_length = -1
    model.setup_caches(1, 32)
    with torch.no_grad():
        model(prompt.view(1, -1), torch.arange(P, device=dev))
        eng = DecodeEngine(model, ths)
        tok = torch.tensor([[17]], device=dev, dtype=torch.int)
        pos = torch.tensor([P], device=dev, dtype=torch.int)
        eng(tok, pos)
        torch.cuda.synchronize()
        inter, dim = eng.inter, eng.dim
        # the state after the full step: layer 1's hand-overs; relaunch layer 1's gate|up-consumer (down) on them
        L = 1
        gu = eng.gu.clone()
        ref = eng.s_down.view(-1)[: dim * 4].clone()
        for _ in range(50):  # alone: reproducible?
            eng._layer(L, tok.data_ptr(), pos.data_ptr(), only=("down",))
            assert torch.equal(eng.s_down.view(-1)[: dim * 4].view(torch.int32), ref.view(torch.int32)), "not reproducible alone"
        tau = float(ths[L]["down"])
        x = (gu[:inter].float() * gu[inter:].float()).half() if eng.gate_act else None
        assert x is not None
        keep = x.float().abs() > tau
        W = model.layers[L].feed_forward.w2.weight  # [dim, inter]
        assert tuple(W.shape) == (dim, inter), W.shape
        print(f"down projection of layer {L}: Z {inter}, N {dim}, kept {int(keep.sum())}, tau {tau:.4f}, gate_act {eng.gate_act}", flush=True)
        ready = f"/tmp/noise_ready_{os.getpid()}"
        child = subprocess.Popen([sys.executable, os.path.join(ROOT, "scripts", "micro", "concurrency_determinism_probe.py"), "--noise-child",
                                  os.environ.get("NOISE", "op_linear_qkv")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                 env={**os.environ, "NOISE_READY_FILE": ready})
        try:
            for _ in range(300):
                if os.path.exists(ready):
                    break
                time.sleep(0.5)
            shown = 0
            for rep in range(2000):
                eng._layer(L, tok.data_ptr(), pos.data_ptr(), only=("down",))
                cur = eng.s_down.view(-1)[: dim * 4]
                if torch.equal(cur.view(torch.int32), ref.view(torch.int32)):
                    continue
                assert torch.equal(eng.gu, gu), "the input changed"
                idx = torch.nonzero(cur.view(torch.int32) != ref.view(torch.int32)).view(-1)
                col, slab = idx // 4, idx % 4
                groups = {}
                for c, s_, i_ in zip(col.tolist(), slab.tolist(), idx.tolist()):
                    groups.setdefault((s_, c // 64, c % 8), []).append((c, i_))
                print(f"repeat {rep}: {idx.numel()} words differ in {len(groups)} (slab, 64-column tile, column mod 8) groups", flush=True)
                for (s_, tile, j), members in list(groups.items())[:4]:
                    cols = torch.tensor([m[0] for m in members], device=dev)
                    d = (cur[cols * 4 + s_] - ref[cols * 4 + s_]).double()
                    Ws = W[cols].double()  # [ncols, inter]
                    a = (d[:, None] * Ws).sum(0) / (Ws * Ws).sum(0)
                    resid = ((d[:, None] - Ws * a[None, :]) ** 2).sum(0).sqrt() / d.norm()
                    best = torch.argsort(resid)[:3]
                    desc = "; ".join(f"row {int(r)} (chunk {int(r) // 64}, chunk mod 4 = {(int(r) // 64) % 4}, kept {bool(keep[r])}) a = {float(a[r]):+.5f} vs x_r = {float(x[r]):+.5f}, residual {float(resid[r]):.2e}"
                                     for r in best)
                    print(f"   slab {s_} tile {tile} column-in-lane {j}: {len(members)} columns, |delta| {float(d.abs().max()):.3e}: {desc}", flush=True)
                shown += 1
                if shown >= 6:
                    break
        finally:
            child.kill()


if __name__ == "__main__":
    main()
