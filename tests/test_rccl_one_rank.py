"""GPU: every RCCL ("nccl") code path of the repo, executed with a ONE-rank process group on the one leased GPU — the product
code of SURVEY 8(f) rank 4 (tensor parallelism) and of bench.py's N > 1 contract that a multi-GPU node would otherwise be the
first to run (reference: gpt-fast/tp.py:36-51,120-121,139-140; gpt-fast/generate.py:249-256).  Each check runs in a process
of its own (a process group is process state)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out):
    return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


def _env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("arch,precision", [("7B", "fp16"), ("llama-3-8b", "bf16")])
def test_engine_graph_holds_the_rccl_allreduce_and_stays_bit_identical(arch, precision):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_worker.py"), arch, precision, "2"],
                         capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    r = _json_line(out.stdout)
    print(json.dumps(r))
    assert r["backend"] == "nccl" and r["world"] == 1
    # the collective is a node of the captured decode step: two per layer were issued while the stream was capturing
    assert r["reduce_calls_while_capturing"] == 2 * r["n_layer"] and r["reduce_calls_while_capturing_presummed"] == 2 * r["n_layer"]
    # the sum over one rank is the identity: tokens of 50 replays, logits, residual stream, every KV row — bit for bit
    assert r["graph_with_allreduce_bit_identical"], r
    assert r["presummed_no_reduce_bit_identical"] and r["presummed_allreduce_bit_identical"], r
    # payloads: the slab buffer [dim][4] fp32 against ONE fp32 [dim]
    dim = 4096
    assert r["reduce_bytes_presummed"] == [dim * 4] and all(b in (dim * 16, dim * 32) for b in r["reduce_bytes_slabs"]), r
    assert r["sync_thresholds_identity"] and r["gather_identity"] and r["calibration_with_gather_identical"], r


def test_bench_distributed_branch_with_one_rccl_rank():
    """bench.py's world > 1 branch — init_process_group("nccl", device_id=...), barrier, the float64 MAX all-reduce on the device —
    taken by one rank (TEAL_BENCH_FORCE_DIST=nccl): same keys as the plain line, n_gpus 1, value within 2 %."""
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10", "--n_layer", "8", "--no-dense",
            "--no-cpu-baseline", "--no-context-sweep", "--no-live-traffic"]
    lines = {}
    for tag, env in (("plain", _env()), ("nccl", _env(TEAL_BENCH_FORCE_DIST="nccl"))):
        out = subprocess.run(args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert out.returncode == 0, (tag, out.stdout[-1500:], out.stderr[-3000:])
        lines[tag] = _json_line(out.stdout)
    a, b = lines["plain"], lines["nccl"]
    assert set(a) == set(b), set(a) ^ set(b)
    assert b["n_gpus"] == 1 and b["config"]["process_group"].startswith("nccl, world 1") and "process_group" not in a["config"]
    assert set(a["config"]) | {"process_group"} == set(b["config"])
    assert abs(b["value"] / a["value"] - 1.0) < 0.02, (a["value"], b["value"])
    assert b["roofline"] is not None and b["roofline"]["kept_fraction"] > 0.3
