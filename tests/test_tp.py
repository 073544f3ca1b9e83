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
