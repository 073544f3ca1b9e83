"""Worker of tests/test_rccl_one_rank.py: a ONE-rank RCCL ("nccl") process group on the one leased GPU, so that every RCCL code
path of the tensor-parallel decode step executes before a multi-GPU node runs it (round-5 verdict, item 1):

  * tp.make_reduce()'s nccl branch, `capturable`: the all-reduce CAPTURED as a node of DecodeEngine.capture_loop's hipGraph
    (two per layer), 50 replays — the sum over one rank is the identity, so tokens, logits and KV rows must be BIT-IDENTICAL to
    the same engine without a reduce;
  * the presummed hand-over (reduce_presummed: wo / down -> TEAL_OUT_SLAB_SUM, one fp32 [dim] per all-reduce), with and without
    the reduce: bit-identical again (the launch's last slice adds the partials in slice order — the consumer's own order);
  * tp.sync_thresholds' device branch and tp.make_gather's device branch: identities on one rank.

Reference: gpt-fast/tp.py:36-51 (process group), :120-121, :139-140 (the two all-reduces per block), gpt-fast/generate.py:249-256.
Prints one JSON line."""
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teal_amd.gpt_fast import generate as G  # noqa: E402
from teal_amd.gpt_fast import tp  # noqa: E402
from teal_amd.gpt_fast.engine import DecodeEngine  # noqa: E402


def main():
    arch, precision = sys.argv[1], sys.argv[2]
    n_layer = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[precision]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = "cuda:0"
    res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "n_layer": n_layer}
    P, STEPS = 6, 50
    model = G.build_synthetic_model(arch, dev, dt, seed=11, n_layer=n_layer)
    ths = G.apply_sparsity(model, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True)
    prompt = torch.randint(0, model.config.vocab_size, (P,), device=dev, dtype=torch.int, generator=torch.Generator(device=dev).manual_seed(2))
    tok0 = torch.tensor([17], device=dev, dtype=torch.int)

    def run(reduce=None, presum=False, timed=False):
        """prompt through the module path, then STEPS decode steps of the fused engine from ONE hipGraph replay each"""
        model.max_seq_length = -1
        model.setup_caches(1, P + STEPS + 8)
        model.tp_reduce = reduce
        with torch.no_grad():
            model(prompt.view(1, -1), torch.arange(P, device=dev))
            eng = DecodeEngine(model, ths, reduce_presummed=presum)
            assert (eng.reduce is reduce) and eng.presum == presum
            eng.manual_seed(99)
            toks = eng.decode_n(tok0, P, STEPS, use_graph=True).tolist()
            assert eng._graph is not None and not getattr(eng, "tp_capture_error", None), getattr(eng, "tp_capture_error", None)
            torch.cuda.synchronize()
            out = {"tokens": toks, "logits": eng.logits.clone(), "resid": [r.clone() for r in eng.resid],
                   "kv": [(l.attention.kv_cache.k_cache[0, :, :P + STEPS].clone(), l.attention.kv_cache.v_cache[0, :, :P + STEPS].clone())
                          for l in model.layers],
                   "reduce_bytes": eng.reduce_bytes(), "slabs": (eng.n_wo.value, eng.n_down.value)}
            if timed:
                g = eng.capture_loop(0.8, 200)
                eng.pos_buf.fill_(P)
                ts = []
                for _ in range(5):
                    eng.pos_buf.fill_(P)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(40):
                        g.replay()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) / 40 * 1e6)
                out["us_per_step"] = sorted(ts)[len(ts) // 2]
        model.tp_reduce = None
        return out

    def same(a, b):
        return (a["tokens"] == b["tokens"] and torch.equal(a["logits"], b["logits"]) and all(torch.equal(x, y) for x, y in zip(a["resid"], b["resid"]))
                and all(torch.equal(ka, kb) and torch.equal(va, vb) for (ka, va), (kb, vb) in zip(a["kv"], b["kv"])))

    calls = {"n": 0, "capturing": 0, "bytes": set()}
    inner = tp.make_reduce()
    assert inner.capturable, "RCCL all-reduce must be capturable"

    def counted(t):
        calls["n"] += 1
        calls["capturing"] += int(torch.cuda.is_current_stream_capturing())
        calls["bytes"].add(t.numel() * t.element_size())
        return inner(t)
    counted.capturable = True

    base = run(None, False, timed=True)
    res["us_per_step_no_reduce"] = base["us_per_step"]
    red = run(counted, False, timed=True)
    res["reduce_calls_host_side"] = calls["n"]            # warm-up + capture (host side runs at capture time only)
    res["reduce_calls_while_capturing"] = calls["capturing"]
    res["reduce_bytes_slabs"] = sorted(calls["bytes"])
    res["graph_with_allreduce_bit_identical"] = same(base, red)
    res["us_per_step_allreduce_slabs"] = red["us_per_step"]
    res["reduce_payload_slabs"] = red["reduce_bytes"]
    calls.update(n=0, capturing=0, bytes=set())
    pre_plain = run(None, True, timed=True)
    res["presummed_no_reduce_bit_identical"] = same(base, pre_plain)
    res["us_per_step_presummed_no_reduce"] = pre_plain["us_per_step"]
    pre = run(counted, True, timed=True)
    res["presummed_allreduce_bit_identical"] = same(base, pre)
    res["reduce_calls_while_capturing_presummed"] = calls["capturing"]
    res["reduce_bytes_presummed"] = sorted(calls["bytes"])
    res["us_per_step_allreduce_presummed"] = pre["us_per_step"]
    res["reduce_payload_presummed"] = pre["reduce_bytes"]
    res["slabs_wo_down"] = list(base["slabs"])
    res["tokens_head"] = base["tokens"][:8]
    # sync_thresholds' device branch (RCCL reduces device tensors only) and make_gather's: identities on one rank
    got = tp.sync_thresholds(ths, model, force=True)
    res["sync_thresholds_identity"] = all(abs(got[i][k] - float(ths[i][k])) == 0.0 for i in range(len(ths)) for k in ths[i])
    v = torch.randn(12345, device=dev)
    res["gather_identity"] = bool(torch.equal(tp.make_gather()(v), v))
    # the calibration with the gather in the loop (engine.calibrate_on_decode through model.tp_gather): same thresholds as without
    model.max_seq_length = -1
    model.setup_caches(1, P + STEPS + 8)
    with torch.no_grad():
        model(prompt.view(1, -1), torch.arange(P, device=dev))
        sp = {p: [0.5] * n_layer for p in DecodeEngine.SITE}
        e0 = DecodeEngine(model, ths)
        a = e0.calibrate_on_decode(sp, tok0, P, 24)
        model.tp_gather = tp.make_gather()
        e1 = DecodeEngine(model, ths)
        assert e1.gather is not None
        b = e1.calibrate_on_decode(sp, tok0, P, 24)
        model.tp_gather = None
    res["calibration_with_gather_identical"] = all(a[i][k] == b[i][k] for i in range(n_layer) for k in a[i])
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
