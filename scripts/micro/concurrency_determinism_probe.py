#!/usr/bin/env python3
"""Is the fused decode step bit-reproducible while ANOTHER PROCESS keeps the same GPU busy?  (tests/test_soak.py shows 5 800
bit-identical hipGraph replays in one process; tests/test_tp_gpu.py's two-ranks-on-one-GPU test showed the UNSHARDED engine's
logits differing in ~1 of 7 runs, round 6.)  One decode step of a 2-layer Llama-2-7B-width model at 50 %, repeated from the same
state; after EVERY launch the buffer that launch completes is compared on the device with the first run's.  Reports the first
differing (repeat, layer, stage, buffer) with the differing elements.

    python scripts/micro/concurrency_determinism_probe.py [--repeats 2000] [--noise matmul|copy|engine|none] [--graph 0|1]
"""
import argparse
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def noise(kind):
    torch.cuda.set_device(0)
    # never outlive the probe (a killed parent — a test timeout — must not leave a process hammering the GPU next to whatever runs next)
    import threading
    parent = os.getppid()

    def _watch():
        while True:
            time.sleep(1.0)
            if os.getppid() != parent:
                os._exit(0)
    threading.Thread(target=_watch, daemon=True).start()
    if kind == "matmul":
        a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
        while True:
            for _ in range(50):
                a @ a
            torch.cuda.synchronize()
    if kind == "copy":
        a = torch.empty(256 << 20, device="cuda", dtype=torch.uint8)
        b = torch.empty_like(a)
        while True:
            for _ in range(20):
                b.copy_(a)
            torch.cuda.synchronize()
    if kind == "malloc":  # hipMalloc / fill / hipFree of large blocks (what torch.cuda.empty_cache() + a rebuild does)
        while True:
            junk = [torch.full(((64 + 32 * k) << 18,), float(k), device="cuda") for k in range(3)]
            torch.cuda.synchronize()
            del junk
            torch.cuda.empty_cache()
    if kind == "h2d":  # pageable host memory copied to the device (staging / pinning inside the runtime)
        while True:
            h = torch.randn(8 << 20)
            d = h.to("cuda")
            torch.cuda.synchronize()
            del h, d
    if kind == "pinned":  # pinned host allocations coming and going
        while True:
            h = torch.empty(16 << 20, pin_memory=True)
            d = h.to("cuda", non_blocking=True)
            torch.cuda.synchronize()
            del h, d
            torch._C._host_emptyCache() if hasattr(torch._C, "_host_emptyCache") else None
    if kind == "build":  # the model build of the probes: random init on the device, thresholds, caches; freed again
        from teal_amd.gpt_fast import generate as G
        while True:
            m = G.build_synthetic_model("7B", "cuda", torch.float16, seed=5, n_layer=2)
            G.apply_sparsity(m, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True, decode_calibration=False)
            torch.cuda.synchronize()
            del m
            torch.cuda.empty_cache()
    if kind in ("randn", "prefill", "quantile", "sdpa", "cat"):  # the pieces of "build", one at a time
        from teal_amd.gpt_fast import generate as G
        m = G.build_synthetic_model("7B", "cuda", torch.float16, seed=5, n_layer=2)
        m.setup_caches(1, 32)
        g = torch.Generator(device="cuda").manual_seed(1)
        toks = torch.randint(0, 32000, (1, 24), device="cuda", dtype=torch.int)
        big = torch.randn(1 << 20, device="cuda")
        q = torch.randn(1, 32, 24, 128, device="cuda", dtype=torch.float16)
        with torch.no_grad():
            while True:
                for _ in range(10):
                    if kind == "randn":
                        w = torch.randn((11008, 4096), device="cuda", dtype=torch.float32, generator=g) * 0.02
                        m.layers[0].feed_forward.w1.weight.data.copy_(w)
                        del w
                    elif kind == "prefill":
                        m(toks, torch.arange(24, device="cuda"))
                    elif kind == "quantile":
                        torch.quantile(big[::5].abs(), 0.5)
                    elif kind == "sdpa":
                        torch.nn.functional.scaled_dot_product_attention(q, q, q, is_causal=True)
                    else:
                        torch.cat([big, big]).abs().float().flatten()
                torch.cuda.synchronize()
                if os.environ.get("NOISE_READY_FILE"):
                    open(os.environ["NOISE_READY_FILE"], "w").write("ready")
    if kind.startswith("op_"):  # single operations of the dense prompt pass (24 tokens, Llama-2-7B widths)
        import torch.nn.functional as F
        T = int(os.environ.get("NOISE_TOKENS", "24"))
        x = torch.randn(1, T, 4096, device="cuda", dtype=torch.float16)
        h = torch.randn(1, T, 11008, device="cuda", dtype=torch.float16)
        w_qkv = torch.randn(12288, 4096, device="cuda", dtype=torch.float16) * 0.02
        w_o = torch.randn(4096, 4096, device="cuda", dtype=torch.float16) * 0.02
        w_gu = torch.randn(11008, 4096, device="cuda", dtype=torch.float16) * 0.02
        w_d = torch.randn(4096, 11008, device="cuda", dtype=torch.float16) * 0.02
        w_head = torch.randn(32000, 4096, device="cuda", dtype=torch.float16) * 0.02
        emb = torch.nn.Embedding(32000, 4096, device="cuda", dtype=torch.float16)
        toks = torch.randint(0, 32000, (1, T), device="cuda")
        nw = torch.ones(4096, device="cuda", dtype=torch.float16)
        with torch.no_grad():
            while True:
                for _ in range(20):
                    if kind == "op_linear_qkv":
                        F.linear(x, w_qkv)
                    elif kind == "op_linear_o":
                        F.linear(x, w_o)
                    elif kind == "op_linear_gu":
                        F.linear(x, w_gu)
                    elif kind == "op_linear_d":
                        F.linear(h, w_d)
                    elif kind == "op_linear_head":
                        F.linear(x, w_head)
                    elif kind == "op_norm":
                        xf = x.float()
                        (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).type_as(x) * nw
                    elif kind == "op_silu":
                        F.silu(h) * h
                    elif kind == "op_emb":
                        emb(toks)
                    elif kind == "op_add":
                        x + x
                torch.cuda.synchronize()
                if os.environ.get("NOISE_READY_FILE"):
                    open(os.environ["NOISE_READY_FILE"], "w").write("ready")
    if kind == "spawn":  # processes that open the device (queues created and destroyed: the scheduler's run list changes)
        while True:
            subprocess.run([sys.executable, "-c", "import torch; torch.zeros(1, device='cuda'); torch.cuda.synchronize()"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if kind == "engine":  # a second decode engine of the same kind: what the two-rank test runs next to each rank
        from teal_amd.gpt_fast import generate as G
        from teal_amd.gpt_fast.engine import DecodeEngine
        m = G.build_synthetic_model("7B", "cuda", torch.float16, seed=5, n_layer=4)
        ths = G.apply_sparsity(m, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True, decode_calibration=False)
        m.max_seq_length = -1
        m.setup_caches(1, 64)
        with torch.no_grad():
            m(torch.randint(0, 32000, (1, 6), device="cuda", dtype=torch.int), torch.arange(6, device="cuda"))
            eng = DecodeEngine(m, ths)
            tok = torch.tensor([[3]], device="cuda", dtype=torch.int)
            pos = torch.tensor([6], device="cuda", dtype=torch.int)
            while True:
                for _ in range(20):
                    eng(tok, pos)
                torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=2000)
    ap.add_argument("--noise", default="engine")
    ap.add_argument("--noise-child", default=None)
    ap.add_argument("--n_layer", type=int, default=2)
    ap.add_argument("--weights", default="16bit", choices=["16bit", "int8", "int4"])
    ap.add_argument("--arch", default="7B", help="model widths of the probed step (7B, llama-3-8b, 70B ...)")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--max_seq", type=int, default=32, help="KV cache rows (grouped-query models switch to the grouped attention kernel from 2048 / 4096)")
    ap.add_argument("--pos", type=int, default=0, help="decode position (default: right behind the 6-token prompt)")
    ap.add_argument("--sparsity", type=float, default=0.5)
    ap.add_argument("--churn", type=int, default=0, help="1: hipMalloc / fill / hipFree of a few large blocks before every step")
    ap.add_argument("--rebuild", type=int, default=0, help="N > 0: build a fresh DecodeEngine every N steps (same model)")
    ap.add_argument("--tag", default="p0")
    a = ap.parse_args()
    if a.noise_child:
        return noise(a.noise_child)
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast.engine import DecodeEngine
    child = None
    if a.noise != "none":
        ready = f"/tmp/noise_ready_{os.getpid()}"
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--noise-child", a.noise], stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL, env={**os.environ, "NOISE_READY_FILE": ready})
        if a.noise.startswith("op_") or a.noise in ("randn", "prefill", "quantile", "sdpa", "cat"):
            for _ in range(300):
                if os.path.exists(ready):
                    break
                time.sleep(0.5)
        else:
            time.sleep(25 if a.noise == "engine" else 8)  # let it get going
    try:
        dev = "cuda"
        model = G.build_synthetic_model(a.arch, dev, torch.float16 if a.precision == "fp16" else torch.bfloat16, seed=11, n_layer=a.n_layer)
        if a.weights == "int8":
            from teal_amd.quantize import quantize_model_int8
            quantize_model_int8(model)
        elif a.weights == "int4":
            from teal_amd.quantize import quantize_model_int4
            quantize_model_int4(model, 32)
        ths = G.apply_sparsity(model, sparsity=a.sparsity, hist_path=None, greedy_lookup=None, synthetic=True, decode_calibration=False)
        P = 6
        prompt = torch.randint(0, 32000, (P,), device=dev, dtype=torch.int, generator=torch.Generator(device=dev).manual_seed(2))
        model.max_seq_length = -1
        model.setup_caches(1, a.max_seq)
        with torch.no_grad():
            model(prompt.view(1, -1), torch.arange(P, device=dev))
            if a.pos > P:  # a long context: fill the cache rows up to the decode position with random rows
                for layer in model.layers:
                    kc = layer.attention.kv_cache
                    kc.k_cache[:, :, P:a.pos].normal_(0, 0.5)
                    kc.v_cache[:, :, P:a.pos].normal_(0, 0.5)
                P = a.pos
            eng = DecodeEngine(model, ths)
            tok = torch.tensor([[17]], device=dev, dtype=torch.int)
            pos = torch.tensor([P], device=dev, dtype=torch.int)

            def bufs(stage, i):
                L = model.layers[i] if i >= 0 else None
                roped = bool(getattr(eng, "rope_epilogue", False)) and eng.n_qkv.value == 0  # the projection's epilogue rotates and appends
                if stage == "qkv":
                    if roped:
                        return {"q": eng.qkv[: eng.qdim], "k_row": L.attention.kv_cache.k_cache[0, :, P], "v_row": L.attention.kv_cache.v_cache[0, :, P],
                                "resid_B": eng.resid[1]}
                    st = (max(1, eng.n_qkv.value) + 3) & ~3
                    return {"s_qkv": eng.s_qkv.view(-1)[: eng.nqkv * st], "resid_B": eng.resid[1]}
                if stage == "attn":
                    if roped:
                        return {"att_ws": eng.att_ws}
                    return {"att_ws": eng.att_ws, "k_row": L.attention.kv_cache.k_cache[0, :, P], "v_row": L.attention.kv_cache.v_cache[0, :, P]}
                if stage == "wo":
                    return {"s_wo": eng.s_wo.view(-1)[: eng.dim * 4]}
                if stage == "gate_up":
                    return {"gu": eng.gu, "resid_A": eng.resid[0]}
                if stage == "down":
                    return {"s_down": eng.s_down.view(-1)[: eng.dim * 4]}
                return {"logits": eng.logits}

            ref, found = {}, []
            nbad = 0
            from collections import Counter
            first_stage, jhist = Counter(), Counter()

            def hook(when, stage, i):
                if when != "after":
                    return
                for name, t in bufs(stage, i).items():
                    key = (i, stage, name)
                    if key not in ref:
                        ref[key] = t.clone()
                    elif not found and not torch.equal(ref[key].view(torch.uint8), t.view(torch.uint8)):
                        d = (ref[key] != t) if ref[key].dtype != torch.float32 else (ref[key].view(torch.int32) != t.view(torch.int32))
                        idx = torch.nonzero(d.view(-1)).view(-1)
                        found.append((key, int(idx.numel()), idx[:12].tolist(), ref[key].reshape(-1)[idx[:6]].tolist(), t.reshape(-1)[idx[:6]].tolist()))
                        first_stage[stage] += 1
                        if name in ("gu", "s_wo", "s_down"):  # which of a lane's eight columns (no rotation mixes neighbours here)
                            col = idx // 4 if name != "gu" else idx
                            for j, c in zip(*[v.tolist() for v in torch.unique(col % 8, return_counts=True)]):
                                jhist[j] += c

            eng(tok, pos, hook=hook)
            torch.cuda.synchronize()
            t0 = time.time()
            for r in range(a.repeats):
                if a.churn:
                    junk = [torch.full(((64 + 32 * k) << 18,), float(k), device=dev) for k in range(3)]  # 64 / 96 / 128 MB
                    del junk
                    torch.cuda.empty_cache()
                if a.rebuild and r % a.rebuild == a.rebuild - 1:
                    del eng
                    torch.cuda.empty_cache()
                    eng = DecodeEngine(model, ths)
                eng(tok, pos, hook=hook)
                if found:
                    torch.cuda.synchronize()
                    key, n, idx, was, now = found[0]
                    if nbad < 5:
                        print(f"[{a.tag}] repeat {r}: layer {key[0]} stage {key[1]} buffer {key[2]}: {n} elements differ; first indices {idx}; was {was}; now {now}")
                    found.clear()
                    nbad += 1
                    # which later buffers differ in this same step is a consequence; restart the comparison from a fresh step
            torch.cuda.synchronize()
            print(f"[{a.tag}] {a.repeats} repeats with noise={a.noise} churn={a.churn} rebuild={a.rebuild}: {nbad} repeats differed; done in {time.time() - t0:.1f} s; first differing stage {dict(first_stage)}; column mod 8 of the differing elements {dict(sorted(jhist.items()))}")
    finally:
        if child is not None:
            child.kill()
            child.wait()
            try:
                os.remove(f"/tmp/noise_ready_{os.getpid()}")
            except OSError:
                pass


if __name__ == "__main__":
    main()
