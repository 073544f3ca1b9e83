#!/usr/bin/env python3
"""The prompt pass's validity range (round-5 verdict, item 5): generate() by the reference's definition — new tokens / wall time
INCLUDING the prompt pass (gpt-fast/generate.py:458,487-496) — at prompt lengths 6, 8, 9, 16, 64, 256, 1024 on Llama-2-7B @ 50 %
under --compile: which pass ran (the hand-fused HIP pass serves 2..16 tokens, teal_amd/gpt_fast/prefill.py; longer prompts take the
patched modules under a per-length hipGraph; --eager is the op-by-op pass), its milliseconds, and the tokens/s of the whole call.
The prefill stays DENSE in all of them (kernels/sparse_gemv.py:271,298).  GPU box, through gpurun:

    python scripts/prefill_vs_prompt_length.py [--synthetic 7B] [--lengths 6,8,9,16,64,256,1024] > gpurun_out/r06_prefill_vs_prompt_length.txt
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
    ap.add_argument("--lengths", default="6,8,9,12,16,64,256,1024")
    ap.add_argument("--n_layer", type=int, default=None)
    a = ap.parse_args()
    dev, dt = "cuda", torch.float16
    from teal_amd import runtime
    from teal_amd.gpt_fast.prefill import FusedPrefill
    runtime.init()
    model = G.build_synthetic_model(a.synthetic, dev, dt, n_layer=a.n_layer)
    ths = G.apply_sparsity(model, sparsity=a.sparsity, hist_path=None, greedy_lookup=None, synthetic=True)
    G.relayout_for_engine(model)
    dec = G.EngineDecoder(model, ths, True, 0.8, 200)
    sync = torch.cuda.synchronize
    graphed = G.GraphedPrefill(model)
    pre = FusedPrefill(model, graph=True, fallback=graphed)  # what `generate.py --compile` installs
    print(f"# {a.synthetic} fp16 @ {a.sparsity:.0%}, {a.max_new_tokens} new tokens per call, --compile (engine loop: one hipGraph replay per token)")
    print("%8s | %-42s %10s | %12s %10s | %14s" % ("prompt", "prompt pass taken", "pass ms", "generate ms", "tok/s", "eager pass ms"))
    base = None
    for T in [int(v) for v in a.lengths.split(",")]:
        if T + a.max_new_tokens > model.config.block_size:
            print(f"{T:8d} | does not fit the model's context ({model.config.block_size}) with {a.max_new_tokens} new tokens")
            continue
        prompt = torch.randint(0, model.config.vocab_size, (T,), device=dev, dtype=torch.int, generator=torch.Generator(device=dev).manual_seed(7 + T))
        for _ in range(2):  # capture + warm-up
            G.generate(model, prompt, a.max_new_tokens, dec, prefill=pre)
        walls = []
        for _ in range(3):
            sync(); t0 = time.perf_counter()
            G.generate(model, prompt, a.max_new_tokens, dec, prefill=pre)
            sync(); walls.append((time.perf_counter() - t0) * 1e3)
        w = sorted(walls)[1]
        ts = []
        for _ in range(3):
            sync(); t0 = time.perf_counter(); pre(prompt); sync(); ts.append((time.perf_counter() - t0) * 1e3)
        t_pre = sorted(ts)[1]
        path = {"hip": "hand-fused HIP pass (hipGraph)", "fallback": "patched modules under a hipGraph"}[pre.used]
        te = []
        for _ in range(2):
            sync(); t0 = time.perf_counter(); model(prompt.view(1, -1), torch.arange(0, T, device=dev)); sync(); te.append((time.perf_counter() - t0) * 1e3)
        if T == 8:
            base = t_pre
        note = "" if base is None or T <= 8 else f"   ({t_pre / base:.1f}x the 8-token pass)"
        print("%8d | %-42s %10.2f | %12.2f %10.1f | %14.2f%s" % (T, path, t_pre, w, a.max_new_tokens / w * 1e3, min(te), note))
    print(f"# graphs held: {len(graphed.graphs)} module-path lengths (LRU, at most {graphed.MAX_GRAPHS}), {len(pre._graphs)} HIP-pass lengths (2..16 only)")


if __name__ == "__main__":
    main()
