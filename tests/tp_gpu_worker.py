"""Worker of tests/test_tp_gpu.py::test_tp_engine_two_ranks_share_one_gpu — launched twice by torch.distributed.run on ONE GPU
(TEAL_TP_BACKEND=gloo: RCCL refuses two ranks on a device; the all-reduce is staged through the host, the decode step runs
eagerly).  Each rank builds the unsharded 2-layer model AND its own shard (generate.build_synthetic_model(shard=tp.apply_tp):
the shard's weights are the slices of the unsharded model's), decodes one token through the fused engine on both, and compares.
Reference: gpt-fast/tp.py:110-140 (what is sharded, where the two all-reduces sit), gpt-fast/generate.py:249-256 (shard before
the weights reach the device)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teal_amd.gpt_fast import generate as G  # noqa: E402
from teal_amd.gpt_fast import tp  # noqa: E402
from teal_amd.gpt_fast.engine import DecodeEngine  # noqa: E402
from teal_amd.monkeypatch import monkeypatch_layer  # noqa: E402


def main():
    arch, precision = sys.argv[1], sys.argv[2]
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[precision]
    rank = tp.maybe_init_dist()
    assert rank is not None and dist.get_backend() == "gloo"
    world = dist.get_world_size()
    dev = f"cuda:{torch.cuda.current_device()}"
    n_layer, P = 2, 6
    res = {"world": world, "n_layer": n_layer}
    prompt = torch.randint(0, 32000, (P,), device=dev, dtype=torch.int, generator=torch.Generator(device=dev).manual_seed(2))
    tok = torch.tensor([[17]], device=dev, dtype=torch.int)
    pos = torch.tensor([P], device=dev, dtype=torch.int)

    def run(model, ths, steps=4):
        """prefill through the module path (S > 1: dense matmuls, the reference's branch), ONE single-token call = the fused
        engine, then a few eager engine steps with the sampler"""
        for i, layer in enumerate(model.layers):
            if not hasattr(layer.attention, "thresh_q"):  # (apply_sparsity patched the unsharded model already)
                monkeypatch_layer(i, layer, 0.0, None, "cuda", thresholds=ths[i])
        model.max_seq_length = -1
        model.setup_caches(1, 32)
        with torch.no_grad():
            pre = model(prompt.view(1, -1), torch.arange(P, device=dev))[0, -1].float().clone()
            logits = model(tok, pos).float().view(-1).clone()  # Transformer.forward -> DecodeEngine (fused_decode)
            eng = model._fused_engine()
            assert isinstance(eng, DecodeEngine)
            eng.manual_seed(99)
            toks = eng.decode_n(tok.view(-1), P, steps, use_graph=False).tolist()
        return pre, logits, eng, toks

    out = {}
    for label, sparsity in (("dense", 0.0), ("sparse", 0.5)):
        full = G.build_synthetic_model(arch, dev, dt, seed=11, n_layer=n_layer)
        # thresholds from the UNSHARDED model's synthetic calibration (deterministic: the same on every rank)
        ths = G.apply_sparsity(full, sparsity=sparsity, hist_path=None, greedy_lookup=None, synthetic=True, decode_calibration=False)
        pre_f, log_f, eng_f, toks_f = run(full, ths)
        assert eng_f.reduce is None
        kc_f = full.layers[1].attention.kv_cache.k_cache[0, :, :P + 1].clone()
        import zlib
        crc = lambda t: zlib.crc32(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes())  # noqa: E731 (bytes: bf16 too)
        torch.cuda.synchronize()
        early = {"decode_full": crc(log_f), "kc_full": crc(kc_f)}
        del eng_f, full
        torch.cuda.empty_cache()
        part = G.build_synthetic_model(arch, dev, dt, seed=11, n_layer=n_layer, shard=tp.apply_tp)
        assert part.tp_world == world and part.fused_decode
        calls = {"n": 0}
        inner = part.tp_reduce

        def counted(t, inner=inner):
            calls["n"] += 1
            return inner(t)
        counted.capturable = False
        part.tp_reduce = counted
        pre_p, log_p, eng_p, toks_p = run(part, ths)
        assert eng_p.reduce is counted and eng_p.qdim * world == eng_p.dim
        out[label] = (pre_f, log_f, pre_p, log_p, toks_f, toks_p)
        # digests of what each side produced (run-to-run reproducibility of the two-process test is checked on these)
        res.setdefault("digest", {})[label] = {"prefill_full": crc(pre_f), "decode_full": crc(log_f), "prefill_rank": crc(pre_p),
                                               "decode_rank": crc(log_p), "kc_full": crc(kc_f),
                                               "early": early,
                                               "ths": zlib.crc32(json.dumps(ths, sort_keys=True).encode()),
                                               "resid_rank": crc(eng_p.resid[0]) ^ crc(eng_p.resid[1]), "gu_rank": crc(eng_p.gu),
                                               "s_wo_rank": crc(eng_p.handover_sum("wo")), "s_down_rank": crc(eng_p.handover_sum("down")),
                                               "qkv_rank": crc(eng_p.qkv), "att_rank": crc(eng_p.att_ws)}
        if label == "dense":
            res["engine_fused"] = True
            res["reduces_per_step"] = calls["n"] // (1 + 4)  # one forward call + four decode_n steps
            nkv = part.config.n_local_heads
            kc_p = part.layers[1].attention.kv_cache.k_cache[0, :, :P + 1]
            # (layer 1's rows: its input went through layer 0's 16-bit all-reduce of rounded partials on the module path — a few
            #  output ulps: 2^-10 relative in fp16, 2^-7 in bf16)
            tol = 2e-2 if dt == torch.float16 else 1e-1
            res["kv_rows_equal"] = bool(torch.allclose(kc_p.float(), kc_f[rank * nkv:(rank + 1) * nkv].float(), atol=tol, rtol=tol))
            res["kv_rows_max_err"] = float((kc_p.float() - kc_f[rank * nkv:(rank + 1) * nkv].float()).abs().max())
        else:
            kf = eng_p.kept_fractions(tok, pos)
            res["kept_o"], res["kept_down"] = kf["o"], kf["down"]
        del eng_p, part
        torch.cuda.empty_cache()
    pre_f, log_f, pre_p, log_p, toks_f, toks_p = out["dense"]
    scale = float(log_f.abs().max())
    res["ulp"] = scale * (2.0 ** -10 if dt == torch.float16 else 2.0 ** -7)
    res["dense_max_err"] = float((log_f - log_p).abs().max())
    res["dense_prefill_max_err"] = float((pre_f - pre_p).abs().max())
    res["tokens_equal_dense"] = toks_f == toks_p
    res["first_token_equal_dense"] = toks_f[0] == toks_p[0]
    _, log_f, _, log_p, _, _ = out["sparse"]
    res["sparse_cosine"] = float(torch.nn.functional.cosine_similarity(log_f, log_p, dim=0))
    # every rank must have seen the same thing
    flags = torch.tensor([res["dense_max_err"], res["sparse_cosine"]], dtype=torch.float64)
    gathered = [torch.zeros_like(flags) for _ in range(world)]
    dist.all_gather(gathered, flags)
    res["dense_max_err"] = float(max(g[0] for g in gathered))
    res["sparse_cosine"] = float(min(g[1] for g in gathered))
    if os.environ.get("TEAL_TP_WORKER_DIGESTS"):  # every rank's digests (run-to-run reproducibility probes)
        print("RANKDIGEST " + json.dumps({"rank": rank, "digest": res.get("digest")}), flush=True)
    if rank == 0:
        print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
