"""PAIR (gate | up in one workgroup, silu * mul + keep masks in the epilogue) against the unpaired stages, tiny model, one
step, every hand-over compared bit for bit (tests/test_engine.py::test_pair_and_masked_stages_equal_unfused_stages)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_engine import _models  # noqa: E402
from teal_amd.gpt_fast.engine import DecodeEngine  # noqa: E402

DEV = "cuda"
from teal_amd import _lib  # noqa: E402
_cm = _lib.diagnostics()
LD = _cm.__enter__()
_, m1, ths = _models(torch.float16, 0.5)
_, m2, _ = _models(torch.float16, 0.5)
for t in ths:
    t["up"] = t["gate"] * 0.8
prompt = torch.tensor([3, 141, 59, 26, 500, 358], device=DEV, dtype=torch.int)
with torch.no_grad():
    for m in (m1, m2):
        m(prompt.view(1, -1), torch.arange(6, device=DEV))
    e1, e2 = DecodeEngine(m1, ths, pair=True), DecodeEngine(m2, ths, pair=False)
    tok = torch.tensor([[5]], device=DEV, dtype=torch.int)
    pos = torch.tensor([6], device=DEV, dtype=torch.int)
    snaps = []
    for e in (e1, e2):
        sn = {}

        def hook(when, stage, i, e=e, sn=sn):
            if when != "after" or i < 0:
                return
            torch.cuda.synchronize()
            if i == 0:
                print(f"   [{'pair' if e is e1 else 'unpaired'}] {stage}: {LD.teal_last_launch_desc().decode()}")
            sn[(i, stage)] = {"resid0": e.resid[0].clone(), "resid1": e.resid[1].clone(), "att_ws": e.att_ws.clone(),
                              "s_wo": e.handover_sum("wo").clone(), "s_down": e.handover_sum("down").clone(), "h_mlp": e.h_mlp.clone(), "gu": e.gu.clone()}
        e(tok, pos, hook=hook)
        snaps.append(sn)
    inter = e1.inter
    for key in snaps[0]:
        a, b = snaps[0][key], snaps[1][key]
        line = []
        for nm in ("resid0", "resid1", "att_ws", "s_wo", "s_down"):
            x, y = a[nm].view(-1), b[nm].view(-1)
            xi = x.view(torch.int32) if x.dtype == torch.float32 else x.view(torch.int16)
            yi = y.view(torch.int32) if y.dtype == torch.float32 else y.view(torch.int16)
            nd = int((xi != yi).sum())
            if nd:
                line.append(f"{nm} {nd}/{x.numel()}")
        if key[1] == "gate_up":
            gu = b["gu"].view(-1)
            h2 = (gu[:inter].float() * gu[inter:2 * inter].float()).half()
            h1 = a["h_mlp"].view(-1)[:inter]
            bad = torch.nonzero(h1.view(torch.int16) != h2.view(torch.int16)).view(-1)
            line.append(f"h: pair epilogue vs round(silu_gate * up) of the unpaired vector: {bad.numel()}/{inter} differ "
                        f"{[(int(j), float(h1[j]), float(h2[j]), float(gu[j]), float(gu[inter + j])) for j in bad[:4]]}")
        print(f"layer {key[0]} after {key[1]}: " + ("; ".join(line) if line else "all equal"))
