import sys, time, torch
kind = sys.argv[1] if len(sys.argv) > 1 else "all"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60
torch.cuda.set_device(0)
a = torch.empty(8192, 8192, device="cuda", dtype=torch.float16)
b = torch.empty(8192, 8192 + 64, device="cuda", dtype=torch.float16)
c = torch.empty(64 << 20, device="cuda", dtype=torch.float32)
t0 = time.time()
while time.time() - t0 < secs:
    for _ in range(10):
        if kind in ("all", "rng"):
            a.normal_(0, 0.02)
            c.uniform_()
        if kind in ("all", "transpose"):
            b[:, :8192] = a.T
        if kind in ("all", "axpy"):
            c.mul_(1.0001).add_(0.5)
        if kind in ("all", "zeros"):
            z = torch.zeros(32 << 20, device="cuda")
            del z
    torch.cuda.synchronize()
