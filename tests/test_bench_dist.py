"""bench.py at N > 1 runs independent replicas (single-batch decode does not shard): the only
collectives are the timing barrier and the max-reduction.  Exercised here with world_size 2 on gloo."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_replicas_time_is_max_over_ranks_and_value_is_aggregate():
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2
    # 20 steps: rank 0 sleeps 2 ms/step, rank 1 4 ms/step -> job time ~ 80 ms (the max), not 40 ms
    assert 0.075 < r["t"] < 0.4, r
    assert abs(r["value"] - 2 * 20 / r["t"]) < 1e-6


def test_bench_cli_contract():
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.model == "7B" and a.sparsity == 0.5
        sys.argv = ["bench.py", "--gpus", "8", "--steps", "50", "--warmup", "5"]
        a = bench.parse()
        assert (a.gpus, a.steps, a.warmup) == (8, 50, 5)
    finally:
        sys.argv = old
