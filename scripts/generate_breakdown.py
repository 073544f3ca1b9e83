#!/usr/bin/env python3
"""Where the wall time of ONE generate() call goes (the reference's tokens/sec definition counts all of it:
gpt-fast/generate.py:458,487-496): prefill (eager module path / hipGraph), first-token sampler, the device-resident decode
loop, the read-back — each bracketed by a synchronise, next to the un-instrumented call.  GPU box, through gpurun:

    python scripts/generate_breakdown.py [--synthetic 7B] [--sparsity 0.5] [--max_new_tokens 200]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teal_amd.gpt_fast import generate as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", default="7B")
    ap.add_argument("--sparsity", type=float, default=0.5)
    ap.add_argument("--max_new_tokens", type=int, default=200)
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--n_layer", type=int, default=None)
    a = ap.parse_args()
    dev = "cuda"
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.precision]
    from teal_amd import runtime
    runtime.init()
    model = G.build_synthetic_model(a.synthetic, dev, dt, n_layer=a.n_layer)
    ths = G.apply_sparsity(model, sparsity=a.sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
    G.relayout_for_engine(model)
    prompt = torch.randint(0, model.config.vocab_size, (6,), device=dev, dtype=torch.int, generator=torch.Generator(device=dev).manual_seed(7))
    dec = G.EngineDecoder(model, ths, True, 0.8, 200)
    sync = torch.cuda.synchronize

    def timed(fn):
        sync(); t0 = time.perf_counter(); r = fn(); sync(); return r, (time.perf_counter() - t0) * 1e3

    from teal_amd.gpt_fast.prefill import FusedPrefill
    for label, pre in (("eager module prefill", None), ("graphed module prefill", G.GraphedPrefill(model)),
                       ("hand-fused HIP prompt pass, graphed", FusedPrefill(model, graph=True))):
        for _ in range(2):  # capture + warm-up
            G.generate(model, prompt, a.max_new_tokens, dec, prefill=pre)
        walls = []
        for _ in range(3):
            _, ms = timed(lambda: G.generate(model, prompt, a.max_new_tokens, dec, prefill=pre))
            walls.append(ms)
        # the same call, piece by piece
        T = prompt.numel()
        _, t_setup = timed(lambda: model.setup_caches(1, min(T + a.max_new_tokens, model.config.block_size)))
        logits, t_pre = timed(lambda: pre(prompt) if pre is not None else model(prompt.view(1, -1), torch.arange(0, T, device=dev)))
        tok, t_samp = timed(lambda: G.sample(logits, temperature=0.8, top_k=200)[0].clone())
        eng = dec.model
        _, t_dec = timed(lambda: eng.decode_n(tok, T, a.max_new_tokens - 1, temperature=0.8, top_k=200, use_graph=True))
        g = eng.capture_loop(0.8, 200)
        eng.tok_buf.copy_(tok.view(1, 1)); eng.pos_buf.fill_(T)

        def replays():
            for _ in range(a.max_new_tokens - 1):
                g.replay()
        _, t_rep = timed(replays)
        w = sorted(walls)[1]
        print(f"[{label}] generate() wall {w:.2f} ms = {a.max_new_tokens / w * 1e3:.1f} tok/s | setup_caches {t_setup:.2f} | prefill {t_pre:.2f} | "
              f"first-token sampler {t_samp:.2f} | decode_n({a.max_new_tokens - 1}) {t_dec:.2f} of which bare replays {t_rep:.2f} "
              f"({t_rep / (a.max_new_tokens - 1) * 1e3:.1f} us/token) | pieces sum {t_setup + t_pre + t_samp + t_dec:.2f}")


if __name__ == "__main__":
    main()
