#!/usr/bin/env python3
"""Phase timing of the decode-attention kernel at bench-like context lengths (diagnostic)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TEAL_LIB_FLAVOR", "diag")  # the tuning / phase-stamp switches exist in libteal_hip_diag.so only
from teal_amd import _lib, runtime
from teal_amd.gpt_fast.model import precompute_freqs_cis

def main():
    L = _lib.load(); runtime.init()
    dt = torch.float16
    n_head = n_kv = 32; hd = 128
    for S, pos in ((464, 200), (464, 450), (2048, 200), (2048, 1500)):
        qkv = torch.randn((n_head + 2 * n_kv) * hd, device="cuda").to(dt)
        kcs = [torch.randn(n_kv, S, hd, device="cuda").to(dt) for _ in range(8)]
        vcs = [torch.randn(n_kv, S, hd, device="cuda").to(dt) for _ in range(8)]
        rope = precompute_freqs_cis(S, hd, 10000, dt).cuda().contiguous()
        y = torch.empty(n_head * hd, device="cuda", dtype=dt)
        m = torch.zeros(n_head * hd // 64, device="cuda", dtype=torch.int64)
        p = torch.tensor([pos], device="cuda", dtype=torch.int32)
        phase = torch.zeros(n_head * 8, dtype=torch.int64, device="cuda")
        rows = []
        for it in range(10):
            phase.zero_(); torch.cuda.synchronize(); L.teal_set_phase_buffer(phase.data_ptr())
            rc = L.teal_decode_attention_masked(qkv.data_ptr(), rope.data_ptr(), p.data_ptr(), kcs[it % 8].data_ptr(), vcs[it % 8].data_ptr(),
                                                y.data_ptr(), m.data_ptr(), 0.1, n_head, n_kv, hd, S, 0, runtime.stream_ptr())
            assert rc == 0
            torch.cuda.synchronize(); L.teal_set_phase_buffer(None)
            if it < 2: continue
            q = phase.view(n_head, 8).cpu().double() * 0.01
            t0 = q[:, 0].min()
            rows.append([float((q[:, 0] - t0).max())] + [float((q[:, i + 1] - q[:, i]).mean()) for i in range(5)] + [float(q[:, 5].max() - t0)])
        r = torch.tensor(rows).median(dim=0).values.tolist()
        print(f"[max_seq {S} pos {pos}] span {r[6]:.2f} us; skew {r[0]:.2f}; loads+rope {r[1]:.2f}; scores {r[2]:.2f}; softmax {r[3]:.2f}; PV {r[4]:.2f}; final {r[5]:.2f}")

if __name__ == "__main__":
    main()
