"""Which kernel instantiation and grid each of the five launches of a decode layer gets, per model width (diagnostics build:
teal_last_launch_desc after every launch of layer 1 of a 2-layer model at 50 %)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from teal_amd import _lib  # noqa: E402
from teal_amd.gpt_fast import generate as G  # noqa: E402
from teal_amd.gpt_fast.engine import DecodeEngine  # noqa: E402

DEV = "cuda"
models = sys.argv[1:] or ["7B", "13B", "30B", "34B", "70B", "llama-3-8b"]
with _lib.diagnostics() as L:
    for name in models:
        dt = torch.bfloat16 if name == "llama-3-8b" else torch.float16
        model = G.build_synthetic_model(name, DEV, dt, seed=11, n_layer=2)
        ths = G.apply_sparsity(model, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True, decode_calibration=False)
        with torch.no_grad():
            model.max_seq_length = -1
            model.setup_caches(1, 64)
            prompt = torch.randint(0, 32000, (6,), device=DEV, dtype=torch.int)
            model(prompt.view(1, -1), torch.arange(6, device=DEV))
            eng = DecodeEngine(model, ths)
            tok = torch.tensor([[17]], device=DEV, dtype=torch.int)
            pos = torch.tensor([6], device=DEV, dtype=torch.int)
            eng(tok, pos)
            out = []

            def hook(when, stage, i):
                if when == "after" and i in (1, -1):
                    torch.cuda.synchronize()
                    out.append(f"   {stage:8s} {L.teal_last_launch_desc().decode()}")
            eng(tok, pos, hook=hook)
        cfg = model.config
        print(f"{name}: dim {cfg.dim} inter {cfg.intermediate_size} heads {cfg.n_head}/{cfg.n_local_heads} pair {eng.pair} att_split {eng.att_split} slabs qkv {eng.n_qkv.value} wo {eng.n_wo.value} down {eng.n_down.value}")
        print("\n".join(out))
        del eng, model
        torch.cuda.empty_cache()
