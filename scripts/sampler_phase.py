#!/usr/bin/env python3
"""Fused sampler: time per call of the multi-workgroup kernel against the single-workgroup window-select kernel
(16 calls on rotating logits inside one hipGraph, as in the decode engine), then the phase stamps of the
single-workgroup kernel (teal_set_phase_buffer).  Benchmark utility (GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TEAL_LIB_FLAVOR", "diag")  # the tuning / phase-stamp switches exist in libteal_hip_diag.so only
from teal_amd import _lib, runtime  # noqa: E402

L = _lib.load()
runtime.init()
WS = runtime.new_workspace(64, 64)  # prepared workspace: scratch of the multi-workgroup sampler
names = ["load + keys + max", "window select", "race over kept", "final reduce"]
for V, dt, code in ((32000, torch.float16, 0), (128256, torch.bfloat16, 1)):
    g = torch.Generator(device="cuda").manual_seed(1)
    state = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")
    tok = torch.zeros(1, dtype=torch.int32, device="cuda")
    lgs = [(torch.randn(V, device="cuda", generator=g) * 2.5).to(dt) for _ in range(16)]
    st = torch.cuda.Stream()
    for multi, label in ((True, "one workgroup per 8192 logits + last arriver"), (False, "single workgroup")):
        with torch.cuda.stream(st):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for lg in lgs:  # (without a prepared workspace the sampler runs as a single workgroup: teal_sample_topk)
                    if multi:
                        assert L.teal_sample_topk_ws(lg.data_ptr(), V, code, 200, 0.8, state.data_ptr(), tok.data_ptr(), None, None, 0, WS.data_ptr(), WS.numel() * 4, st.cuda_stream) == 0
                    else:
                        assert L.teal_sample_topk(lg.data_ptr(), V, code, 200, 0.8, state.data_ptr(), tok.data_ptr(), None, None, 0, st.cuda_stream) == 0
            gr.replay(); st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(8): gr.replay()
            e1.record(st); st.synchronize()
        print(f"V={V}: {label:46s} {e0.elapsed_time(e1) * 1e3 / (8 * 16):6.2f} us per call (incl. launch boundary)")
    g = torch.Generator(device="cuda").manual_seed(1)
    state = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")
    tok = torch.zeros(1, dtype=torch.int32, device="cuda")
    phase = torch.zeros(64, dtype=torch.int64, device="cuda")
    rows = []
    for it in range(12):
        logits = (torch.randn(V, device="cuda", generator=g) * 2.5).to(dt)  # fresh logits: cold in L2 like after lm_head
        torch.cuda.synchronize()
        L.teal_set_phase_buffer(phase.data_ptr())
        assert L.teal_sample_topk(logits.data_ptr(), V, code, 200, 0.8, state.data_ptr(), tok.data_ptr(), None, None, 0, runtime.stream_ptr()) == 0
        torch.cuda.synchronize()
        L.teal_set_phase_buffer(None)
        p = phase[:5].cpu().double() * 10.0 / 1e3
        rows.append([(p[i + 1] - p[i]).item() for i in range(4)])
    med = torch.tensor(rows[2:]).median(dim=0).values.tolist()
    print(f"V={V}: total {sum(med):.2f} us")
    for n, v in zip(names, med):
        print(f"    {n:28s} {v:6.2f} us")
