#!/usr/bin/env python3
"""RCCL with ONE rank on the one leased GPU: init_process_group("nccl", device_id=...), barrier, an fp64 MAX all-reduce, an fp32
SUM all-reduce inside a captured hipGraph (replayed), a device-side all_gather — the calls tests/test_rccl_one_rank.py and the
tensor-parallel decode step rely on (round-5 verdict, item 1).

The capture is tried in BOTH stream-capture error modes, each in a process of its own:
  * global (torch.cuda.graph's default): ProcessGroupNCCL's watchdog thread polls the completion events of earlier collectives
    every 100 ms; while a global-mode capture is in progress on ANY stream that hipEventQuery is refused ("operation not
    permitted when stream is capturing") and the watchdog aborts the process.  A race: the child keeps the capture open for
    ~0.5 s so that a watchdog wake-up falls into it.
  * thread_local (what teal_amd.runtime.graph_capture selects whenever a process group is alive): captures and replays.
Prints one block per mode; exit code 0 if thread_local works."""
import os
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child(mode):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    t0 = time.time()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    print(f"init_process_group(nccl, world 1): ok in {time.time() - t0:.2f} s, backend {dist.get_backend()}", flush=True)
    dist.barrier()
    torch.cuda.synchronize()
    print("barrier: ok", flush=True)
    t = torch.tensor([3.25], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    print("fp64 MAX all_reduce:", float(t.item()), flush=True)
    x = torch.arange(4096 * 4, device="cuda", dtype=torch.float32)
    ref = x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dist.all_reduce(x, op=dist.ReduceOp.SUM)  # eager warm-up: a work object the watchdog will poll
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print("fp32 SUM all_reduce eager: identity", bool(torch.equal(x, ref)), flush=True)
    dist.all_reduce(x, op=dist.ReduceOp.SUM)      # one more, right before the capture
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        x.mul_(2.0)
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        x.add_(1.0)
        t_end = time.time() + 0.5                 # keep the capture open across several watchdog wake-ups
        while time.time() < t_end:
            time.sleep(0.01)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    want = ref.clone()
    for _ in range(5):
        want = want * 2.0 + 1.0
    print(f"all_reduce captured in a hipGraph ({mode} capture mode), 5 replays: equal", bool(torch.equal(x, want)), flush=True)
    ts = []
    for _ in range(50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"graph replay (mul + all_reduce 64 KB + add): median {ts[len(ts) // 2]:.1f} us", flush=True)
    src = torch.randn(1000, device="cuda")
    parts = [torch.empty(1000, device="cuda")]
    dist.all_gather(parts, src)
    print("all_gather: equal", bool(torch.equal(parts[0], src)), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("destroy_process_group: ok", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    rcs = {}
    for mode in ("thread_local", "global"):
        print(f"==== capture_error_mode = {mode}", flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode], capture_output=True, text=True, timeout=240)
        rcs[mode] = r.returncode
        print(r.stdout, end="")
        err = [ln for ln in r.stderr.splitlines() if "capturing" in ln or "what()" in ln or "Error" in ln][:4]
        print(f"exit code {r.returncode}" + ("".join("\n  stderr: " + ln[:220] for ln in err) if r.returncode else ""), flush=True)
    print(f"==== summary: thread_local rc {rcs['thread_local']}, global rc {rcs['global']} "
          f"({'the watchdog aborted the process' if rcs['global'] else 'survived this time (a race against the 100 ms watchdog wake-up)'})")
    return 0 if rcs["thread_local"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
