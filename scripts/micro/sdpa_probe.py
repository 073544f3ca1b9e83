import torch, torch.nn.functional as F, sys
from torch.nn.attention import sdpa_kernel, SDPBackend
dev = "cuda"
def run(S, L, H, backend):
    q = torch.randn(1, H, S, 128, device=dev, dtype=torch.float16)
    k = torch.randn(1, H, L, 128, device=dev, dtype=torch.float16)
    v = torch.randn(1, H, L, 128, device=dev, dtype=torch.float16)
    mask = torch.tril(torch.ones(L, L, dtype=torch.bool, device=dev))[None, None, torch.arange(S, device=dev)]
    with sdpa_kernel(backend):
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0)
    torch.cuda.synchronize()
    print("ok", S, L, H, backend, float(y.float().abs().mean()), flush=True)
which = sys.argv[1]
S, L = int(sys.argv[2]), int(sys.argv[3])
b = {"math": SDPBackend.MATH, "eff": SDPBackend.EFFICIENT_ATTENTION, "flash": SDPBackend.FLASH_ATTENTION}[which]
run(S, L, 32, b)
