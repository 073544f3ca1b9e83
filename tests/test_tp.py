"""Tensor parallelism, host side (second half; reference gpt-fast/tp.py:110-140): wqkv / w1 / w3
column-wise, wo / w2 row-wise, one all-reduce of [1, 1, dim] per attention and per MLP.  World size 2 on gloo, CPU only — the
TEAL launches of each rank are restated with the oracle (the per-rank op is the same sparse GEMV on a narrower image)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def tp_result():
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "tp_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_sharded_model_equals_unsharded(tp_result):
    r = tp_result
    assert r["world"] == 2
    # wqkv [512 -> 256 rows: 4 of 8 query heads + 2 + 2 of 4 KV heads], wo [256, 128], w1 [256, 256], w2 [256, 256]
    assert r["local_shapes"] == [[256, 256], [256, 128], [256, 256], [256, 256], 4, 2]
    assert r["kv_cache_heads"] == 2
    assert r["prefill_max_err"] <= 1e-4 * max(1.0, r["prefill_scale"]), r
    assert r["decode_max_err"] <= 1e-4 * max(1.0, r["prefill_scale"]), r
    assert r["collectives"] == {"all_reduce_calls": 4, "elements_each": 256, "bytes_each": 512, "bytes_per_token": 2048}
    assert r["sync_sharded"] == [{"o": 1.5, "q": 0.5}] and r["sync_replica"] == [{"o": 1.0, "q": 0.5}]  # (rank 0 reports)


def test_teal_masks_and_sums_under_tp(tp_result):
    r = tp_result
    assert r["colwise_exact"] and r["qkv_exact"], "a column-wise shard is the same sparse GEMV on the rank's columns"
    assert r["h_slice_exact"] and r["mask_is_slice"], "|x_local| > tau is the rank-local slice of the global keep mask"
    assert 0.2 < r["kept_down"] < 0.8
    assert r["rowwise_sum_max_err"] <= 1e-9 * max(1.0, r["rowwise_scale"]), r  # fp64 partial sums, one all-reduce


def test_shard_ranges():
    sys.path.insert(0, ROOT)
    from teal_amd.gpt_fast import tp
    assert tp.shard_features(12, 1, 2) == [(6, 12)]
    assert tp.shard_features(4096 + 2 * 1024, 1, 4, [4096, 1024, 1024]) == [(1024, 2048), (4096 + 256, 4096 + 512), (5120 + 256, 5120 + 512)]
    with pytest.raises(ValueError):
        tp.shard_range(10, 0, 4)
    with pytest.raises(ValueError):
        tp.shard_features(10, 0, 2, [4, 4])


def test_int8_weight_only_linears_shard_with_their_scales():
    """column-wise: the rank's output features AND their per-channel scales; row-wise: input features, every scale kept
    (gpt-fast/tp.py:55-106 shards `scales` along with the weight of a WeightOnlyInt8Linear).  The sharded linears applied to
    the rank's inputs reproduce the unsharded int8 linear: concatenation (column-wise), sum over ranks (row-wise)."""
    import torch
    sys.path.insert(0, ROOT)
    from teal_amd.gpt_fast import tp
    from teal_amd.quantize import WeightOnlyInt8Linear
    g = torch.Generator().manual_seed(3)
    lin = torch.nn.Linear(64, 96, bias=False)
    lin.weight.data = torch.randn(96, 64, generator=g)
    x = torch.randn(1, 1, 64, generator=g)
    full = WeightOnlyInt8Linear.from_linear(lin)
    want = full(x)
    cols, rows = [], []
    for rank in range(2):
        c = WeightOnlyInt8Linear.from_linear(lin)
        tp.shard_linear(c, "colwise", rank, 2, [32, 32, 32])  # q | k | v thirds: the rank's half of each
        assert tuple(c.weight.shape) == (48, 64) and tuple(c.scales.shape) == (48,) and c.out_features == 48
        cols.append(c(x))
        r = WeightOnlyInt8Linear.from_linear(lin)
        tp.shard_linear(r, "rowwise", rank, 2)
        assert tuple(r.weight.shape) == (96, 32) and tuple(r.scales.shape) == (96,) and r.in_features == 32
        rows.append(r(x[..., rank * 32:(rank + 1) * 32]))
    got_c = torch.cat([torch.cat([cols[0][..., 16 * p:16 * (p + 1)], cols[1][..., 16 * p:16 * (p + 1)]], dim=-1) for p in range(3)], dim=-1)
    assert torch.equal(got_c, want)
    assert torch.allclose(rows[0] + rows[1], want, rtol=1e-5, atol=1e-5)


def test_bench_runs_exactly_the_requested_steps_through_multi_token_graphs(monkeypatch):
    """bench.py's timed region hands the step count to the engine stepper's run(n): n decode steps whatever the number of tokens a
    graph replay spans (remainder one token per replay), never past the cache."""
    sys.path.insert(0, ROOT)
    import bench

    class Stepper:
        def __init__(self, U, max_seq):
            self.U, self.max_seq, self.pos, self.tokens, self.replays = U, max_seq, 6, 0, []

        def __call__(self):
            if self.pos + 1 >= self.max_seq:
                self.pos = 6
            self.pos += 1
            self.tokens += 1
            self.replays.append(1)

        def run(self, n):  # the loop of engine.make_engine_stepper, on counters
            while n > 0:
                if self.U > 1 and n >= self.U and self.pos + self.U < self.max_seq:
                    self.pos += self.U
                    self.tokens += self.U
                    self.replays.append(self.U)
                    n -= self.U
                else:
                    self()
                    n -= 1

    monkeypatch.setattr(bench, "_sync", lambda: None)
    for U, steps, warm in ((1, 20, 5), (4, 20, 5), (8, 203, 7), (4, 3, 0)):
        s = Stepper(U, 64)
        bench.timed_decode(s, steps, warm, 1)
        assert s.tokens == steps + warm and s.pos < 64
        assert max(s.replays) <= U


def test_synthetic_builder_materialises_each_ranks_slices():
    """generate.build_synthetic_model(shard=apply_tp): the shard happens on the META model, every rank fills only its slices,
    and they ARE the slices of the unsharded synthetic model (same seed, same draw order) — what lets the GPU test compare a
    tensor-parallel engine with the unsharded one."""
    import torch
    sys.path.insert(0, ROOT)
    from teal_amd.gpt_fast import generate as G
    from teal_amd.gpt_fast import tp
    full = G.build_synthetic_model("tiny-gqa-test", "cpu", torch.float32, seed=5)
    cfg = full.config
    q, kv = cfg.n_head * cfg.head_dim, cfg.n_local_heads * cfg.head_dim
    parts = [G.build_synthetic_model("tiny-gqa-test", "cpu", torch.float32, seed=5, shard=lambda m, r=r: tp.apply_tp(m, r, 2)) for r in range(2)]
    for r, m in enumerate(parts):
        assert m.tp_world == 2 and m.tp_rank == r and m.tp_reduce is None and m.config.n_head == cfg.n_head // 2
        assert torch.equal(m.tok_embeddings.weight, full.tok_embeddings.weight) and torch.equal(m.output.weight, full.output.weight)
        for lf, lm in zip(full.layers, m.layers):
            rows = [i for lo, hi in tp.shard_features(q + 2 * kv, r, 2, [q, kv, kv]) for i in range(lo, hi)]
            assert torch.equal(lm.attention.wqkv.weight, lf.attention.wqkv.weight[rows])
            lo, hi = tp.shard_range(q, r, 2)
            assert torch.equal(lm.attention.wo.weight, lf.attention.wo.weight[:, lo:hi])
            lo, hi = tp.shard_range(cfg.intermediate_size, r, 2)
            assert torch.equal(lm.feed_forward.w1.weight, lf.feed_forward.w1.weight[lo:hi])
            assert torch.equal(lm.feed_forward.w3.weight, lf.feed_forward.w3.weight[lo:hi])
            assert torch.equal(lm.feed_forward.w2.weight, lf.feed_forward.w2.weight[:, lo:hi])
