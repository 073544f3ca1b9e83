#!/usr/bin/env python3
"""Run-to-run reproducibility of the FIRST fused decode step after an engine is built (round 6: the two-ranks-on-one-GPU test saw
the unsharded engine's logits differ in ~8 % of its runs).  Each iteration: build the 2-layer Llama-2-7B-width model (fixed seed),
thresholds, prompt pass through the modules, DecodeEngine, ONE step + 4 more; digests of every hand-over buffer after the first
step.  --procs 2 runs two such loops concurrently on the GPU."""
import argparse
import os
import subprocess
import sys
import zlib

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def loop(a, tag):
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine
    if a.fast == 0:
        from teal_amd import _lib
        _lib.load().teal_set_fast(0)
    dev = "cuda"
    P = 6
    crc = lambda t: zlib.crc32(t.detach().contiguous().cpu().numpy().tobytes())  # noqa: E731
    seen, votes = {}, {}
    for it in range(a.iters):
        model = G.build_synthetic_model("7B", dev, torch.float16, seed=11, n_layer=2)
        ths = G.apply_sparsity(model, sparsity=a.sparsity, hist_path=None, greedy_lookup=None, synthetic=True, decode_calibration=False)
        prompt = torch.randint(0, 32000, (P,), device=dev, dtype=torch.int, generator=torch.Generator(device=dev).manual_seed(2))
        tok = torch.tensor([[17]], device=dev, dtype=torch.int)
        pos = torch.tensor([P], device=dev, dtype=torch.int)
        model.max_seq_length = -1
        model.setup_caches(1, 32)
        with torch.no_grad():
            pre = model(prompt.view(1, -1), torch.arange(P, device=dev))[0, -1].float().clone()
            if a.sync_before:
                torch.cuda.synchronize()
            eng = DecodeEngine(model, ths)
            if a.sync_after_build:
                torch.cuda.synchronize()
            stages = {}

            def hook(when, stage, i):
                if when == "after":
                    if stage == "qkv":
                        stages[(i, "q")] = eng.qkv[: eng.qdim].clone()
                        stages[(i, "resid_B")] = eng.resid[1].clone()
                        stages[(i, "k_row")] = model.layers[i].attention.kv_cache.k_cache[0, :, P].clone()
                    elif stage == "attn":
                        stages[(i, "att_ws")] = eng.att_ws.clone()
                    elif stage == "wo":
                        stages[(i, "s_wo")] = eng.s_wo.view(-1)[: eng.dim * 4].clone()
                    elif stage == "gate_up":
                        stages[(i, "gu")] = eng.gu.clone()
                        stages[(i, "resid_A")] = eng.resid[0].clone()
                    elif stage == "down":
                        stages[(i, "s_down")] = eng.s_down.view(-1)[: eng.dim * 4].clone()
                    else:
                        stages[(i, "logits")] = eng.logits.clone()

            eng(tok, pos, hook=hook if a.hook else None)
            logits = eng.logits.clone()
            torch.cuda.synchronize()
            d = {"prefill": crc(pre), "logits": crc(logits)}
            d.update({f"L{k[0]}.{k[1]}": crc(v) for k, v in stages.items()})
            tens = {f"L{k[0]}.{k[1]}": v.detach().cpu() for k, v in stages.items()}
            tens["logits"] = logits.cpu()
        first_bad = None
        for k, v in sorted(d.items(), key=lambda kv: (not kv[0].startswith("L") or kv[0].startswith("L-1"), 0)):  # launch order first
            if k in seen and seen[k][0] != v:
                first_bad = first_bad or k
            seen.setdefault(k, (v, tens.get(k)))
        # majority vote: once three iterations agree on a digest, that one is the reference (iteration 0 may be the odd one)
        votes.setdefault("n", {})
        for k, v in d.items():
            votes["n"].setdefault(k, {}).setdefault(v, [0, tens.get(k)])[0] += 1
        if it >= 3:
            for k in d:
                best = max(votes["n"][k].items(), key=lambda kv: kv[1][0])
                seen[k] = (best[0], best[1][1])
        if first_bad:
            info = ""
            ref_t, cur_t = seen[first_bad][1], tens.get(first_bad)
            if ref_t is not None and cur_t is not None:
                rb, cb = ref_t.contiguous().view(torch.uint8).view(-1), cur_t.contiguous().view(torch.uint8).view(-1)
                es = ref_t.element_size()
                bad = torch.nonzero((rb != cb).view(-1, es).any(1)).view(-1)
                rf, cf = ref_t.float().view(-1), cur_t.float().view(-1)
                info = (f" | {bad.numel()} of {rf.numel()} elements differ; index range {int(bad.min())}..{int(bad.max())}; first {bad[:16].tolist()}; "
                        f"max |diff| {float((rf - cf).abs().max()):.3e} (|ref| max {float(rf.abs().max()):.3e}); ref {rf[bad[:4]].tolist()} now {cf[bad[:4]].tolist()}")
            print(f"[{tag}] iter {it}: DIFFERS first at {first_bad}; all differing: {[k for k, v in d.items() if seen[k][0] != v]}{info}", flush=True)
        else:
            print(f"[{tag}] iter {it}: same as the first iteration ({len(d)} digests)", flush=True)
        del eng, model
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--sparsity", type=float, default=0.5)
    ap.add_argument("--hook", type=int, default=1)
    ap.add_argument("--sync-before", dest="sync_before", type=int, default=0)
    ap.add_argument("--sync-after-build", dest="sync_after_build", type=int, default=0)
    ap.add_argument("--child", default=None)
    ap.add_argument("--fast", type=int, default=1, help="0: force the general kernel (diagnostics build)")
    ap.add_argument("--setprio", type=int, default=1)
    a = ap.parse_args()
    if a.fast == 0:
        os.environ["TEAL_LIB_FLAVOR"] = "diag"
    if a.child or a.procs == 1:
        return loop(a, a.child or "p0")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--iters", str(a.iters), "--sparsity", str(a.sparsity), "--hook", str(a.hook),
                            "--sync-before", str(a.sync_before), "--sync-after-build", str(a.sync_after_build), "--fast", str(a.fast), "--child", f"p{i}"]) for i in range(a.procs)]
    for p in ps:
        p.wait()


if __name__ == "__main__":
    main()
