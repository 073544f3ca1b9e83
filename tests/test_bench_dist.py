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


def test_floor_model_recovers_fixed_cost_and_stream_rate():
    """bench.floor_model fits t = fixed + bytes / rate per launch type through the sparse and the dense point; on numbers made
    from a known model (4 launches, 5.5 us fixed each, 7.7 MB/us, 4.4 us attention) it returns that model and its layer ratio."""
    import bench
    fixed, rate, att = 5.5, 7.7e6, 4.4
    dense_mb = {"qkv": 100.7e6, "wo": 33.6e6, "gate_up": 180.4e6, "down": 90.2e6}
    mk = lambda frac: dict({k: fixed + frac * v / rate for k, v in dense_mb.items()}, attn=att,  # noqa: E731
                           bytes={k: frac * v for k, v in dense_mb.items()})
    s, d = mk(0.5), mk(1.0)
    s["layer"] = sum(s[k] for k in dense_mb) + att
    d["layer"] = sum(d[k] for k in dense_mb) + att
    m = bench.floor_model(s, d, 32)
    for k in dense_mb:
        assert abs(m["launch"][k]["fixed_us"] - fixed) < 0.02 and abs(m["launch"][k]["stream_TBps"] - 7.7) < 0.02
    assert abs(m["fixed_us_per_layer"] - 4 * fixed) < 0.05
    assert abs(m["layer_ratio"]["measured"] - d["layer"] / s["layer"]) < 1e-3
    assert abs(m["layer_ratio"]["model (fixed + attention + streaming)"] - m["layer_ratio"]["measured"]) < 2e-3
    assert abs(m["layer_ratio"]["if the fixed per-launch cost and the attention launch were free"] - 2.0) < 1e-2


def test_forced_one_rank_group_takes_the_distributed_branch():
    """TEAL_BENCH_FORCE_DIST makes a single process take bench.py's N > 1 branch (process group, barrier, max over ranks of the
    timing): here with gloo on CPU; tests/test_rccl_one_rank.py does the same with RCCL on the GPU."""
    code = ("import os, sys, time, json; sys.path.insert(0, %r); import bench\n"
            "rank, world, _ = bench.dist_setup(1)\n"
            "import torch.distributed as dist\n"
            "assert dist.is_initialized() and dist.get_world_size() == 1 and bench._dist_on()\n"
            "t = bench.timed_decode(lambda: time.sleep(0.002), steps=10, warmup=1, world=world)\n"
            "print(json.dumps({'t': t, 'value': bench.aggregate_tokens_per_sec(world, 10, t)}))\n"
            "dist.destroy_process_group()\n" % ROOT)
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", TEAL_BENCH_FORCE_DIST="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert 0.019 < r["t"] < 0.2 and abs(r["value"] - 10 / r["t"]) < 1e-6


def test_roofline_object_follows_survey_8d_to_the_letter():
    """bench.gateup_launch_bytes / roofline_fields: algorithmic bytes of the dominant launch = kept rows of both matrices + Z*2 +
    N_out*2 — nothing of what the fused producer reads — and frac = algorithmic / us / 8e6, on a made-up launch."""
    import bench
    Z, N = 4096, 11008
    b = bench.gateup_launch_bytes(2036, 2040, Z, N, nslabs=4)
    assert b["algorithmic"] == (2036 + 2040) * N * 2 + Z * 2 + 2 * N * 2
    assert b["producer"] == Z * 2 + 4 * Z * 4 + Z * 2
    assert abs(b["kept_fraction"] - (2036 + 2040) / (2 * Z)) < 1e-12
    p = bench.gateup_launch_bytes(2036, 2040, Z, N, nslabs=4, pair=True)
    assert p["algorithmic"] == (2036 + 2040) * N * 2 + Z * 2 + N * 2 and p["producer"] == b["producer"] + N // 8
    r = bench.roofline_fields(b["algorithmic"], 18.29)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - b["algorithmic"] / 18.29 / 8e6) < 1e-9 and abs(r["achieved"] - r["frac"] * 8000.0) < 1e-6
    assert abs(r["frac"] - 0.6135) < 2e-3  # the round-5 launch: 89.8 MB in 18.29 us


def test_floor_model_without_a_slope_stays_valid_json():
    """a dense stage that is not slower than the sparse one (--sparsity 0, or noise on the narrow wo launch) has no fit: the launch
    reports stream_TBps null and the line is still strict JSON"""
    import bench
    mk = lambda f: dict({k: 5.0 + f * v for k, v in (("qkv", 10.0), ("wo", 3.0), ("gate_up", 20.0), ("down", 9.0))}, attn=4.0,  # noqa: E731
                        bytes={k: f * v * 1e6 for k, v in (("qkv", 100.0), ("wo", 33.0), ("gate_up", 180.0), ("down", 90.0))})
    s, d = mk(0.5), mk(1.0)
    d["wo"] = s["wo"] - 0.01  # noise: "dense" faster than sparse
    s["layer"], d["layer"] = 50.0, 77.0
    m = bench.floor_model(s, d, 32)
    assert m["launch"]["wo"]["stream_TBps"] is None and m["launch"]["wo"]["fixed_us"] == round(s["wo"], 2)
    assert m["launch"]["qkv"]["stream_TBps"] is not None
    line = json.dumps(bench._finite({"floor_model": m, "x": float("nan"), "y": [1.0, float("inf")]}), allow_nan=False)
    back = json.loads(line)
    assert back["x"] is None and back["y"] == [1.0, None]
    same = bench.floor_model(s, dict(s, layer=50.0), 32)  # --sparsity 0: both engines keep every row
    json.dumps(bench._finite(same), allow_nan=False)
