#!/usr/bin/env python3
"""Does RCCL work with ONE rank on the one leased GPU — init_process_group("nccl", device_id=...), barrier, an fp64 MAX
all-reduce, an fp32 SUM all-reduce inside a captured hipGraph (replayed), a device-side all_gather?  Prints one line per
step; the TP / bench tests of tests/test_rccl_one_rank.py rely on exactly these calls (round-5 verdict, item 1)."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    t0 = time.time()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    print(f"init_process_group(nccl, world 1): ok in {time.time() - t0:.2f} s, backend {dist.get_backend()}")
    dist.barrier()
    torch.cuda.synchronize()
    print("barrier: ok")
    t = torch.tensor([3.25], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    print("fp64 MAX all_reduce:", float(t.item()))
    x = torch.arange(4096 * 4, device="cuda", dtype=torch.float32)
    ref = x.clone()
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    print("fp32 SUM all_reduce eager: identity", bool(torch.equal(x, ref)))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            x.mul_(2.0)
            dist.all_reduce(x, op=dist.ReduceOp.SUM)
            x.add_(1.0)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        want = ref.clone()
        for _ in range(5):
            want = want * 2.0 + 1.0
        print("all_reduce captured in a hipGraph, 5 replays: equal", bool(torch.equal(x, want)))
        ts = []
        for _ in range(50):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"graph replay (mul + all_reduce 64 KB + add): median {ts[len(ts) // 2]:.1f} us")
    except Exception as e:  # noqa: BLE001
        print("graph capture of all_reduce FAILED:", type(e).__name__, str(e)[:300])
    parts = [torch.empty(1000, device="cuda") for _ in range(1)]
    src = torch.randn(1000, device="cuda")
    dist.all_gather(parts, src)
    print("all_gather: equal", bool(torch.equal(parts[0], src)))
    dist.destroy_process_group()
    print("destroy_process_group: ok")


if __name__ == "__main__":
    sys.exit(main())
