"""Worker of tests/test_bench_dist.py: exercises bench.py's replica timing logic on CPU (gloo)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    rank, world, _ = bench.dist_setup(2)
    per_step = 0.002 * (1 + rank)  # rank 1 is twice as slow: the job time must be rank 1's

    def step():
        time.sleep(per_step)

    t = bench.timed_decode(step, steps=20, warmup=2, world=world)
    if rank == 0:
        print(json.dumps({"world": world, "t": t, "value": bench.aggregate_tokens_per_sec(world, 20, t)}))
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
