"""Which kernels the aggressor operations of concurrency_determinism_probe.py launch (run under rocprofv3 --kernel-trace)."""
import sys, torch, torch.nn.functional as F
kind = sys.argv[1]
T = 24
x = torch.randn(1, T, 4096, device="cuda", dtype=torch.float16)
h = torch.randn(1, T, 11008, device="cuda", dtype=torch.float16)
w = {"qkv": (12288, 4096), "o": (4096, 4096), "gu": (11008, 4096), "d": (4096, 11008), "head": (32000, 4096)}
torch.cuda.synchronize()
if kind in w:
    W = torch.randn(*w[kind], device="cuda", dtype=torch.float16) * 0.02
    torch.cuda.synchronize()
    for _ in range(3):
        F.linear(h if kind == "d" else x, W)
elif kind == "matmul":
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
    for _ in range(3):
        a @ a
elif kind == "sdpa":
    q = torch.randn(1, 32, 24, 128, device="cuda", dtype=torch.float16)
    for _ in range(3):
        F.scaled_dot_product_attention(q, q, q, is_causal=True)
torch.cuda.synchronize()
