"""One decode step of a 1-layer Llama-2-7B-width model through the lean and through the general kernels (diagnostics build),
eager, every hand-over compared bit for bit: where do the two — specified to be bit-identical — part ways?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from teal_amd import _lib  # noqa: E402
from teal_amd.gpt_fast import generate as G  # noqa: E402
from teal_amd.gpt_fast.engine import DecodeEngine  # noqa: E402

DEV = "cuda"


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "7B"
    dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float16
    with _lib.diagnostics() as L:
        model = G.build_synthetic_model(arch, DEV, dt, seed=29, n_layer=2)
        ths = G.apply_sparsity(model, sparsity=0.5, hist_path=None, greedy_lookup=None, synthetic=True)
        NP = 120
        prompt = torch.randint(0, model.config.vocab_size, (NP,), device=DEV, dtype=torch.int, generator=torch.Generator(device=DEV).manual_seed(5))
        with torch.no_grad():
            model.max_seq_length = -1
            model.setup_caches(1, NP + 8)
            model(prompt.view(1, -1), torch.arange(NP, device=DEV))
            eng = DecodeEngine(model, ths)
            tok = torch.tensor([[23]], device=DEV, dtype=torch.int)
            pos = torch.tensor([NP], device=DEV, dtype=torch.int)
            eng(tok, pos)
            torch.cuda.synchronize()
            got = {}
            for fast in (1, 0):
                L.teal_set_fast(fast)
                snaps = {}

                def hook(when, stage, i, snaps=snaps):
                    if when != "after" or i < 0:
                        return
                    torch.cuda.synchronize()
                    at = model.layers[i].attention
                    kc, vc = at.kv_cache.k_cache, at.kv_cache.v_cache
                    d = {"k_row": kc[0, :, NP].clone(), "v_row": vc[0, :, NP].clone(), "att_ws": eng.att_ws.clone(), "resid0": eng.resid[0].clone(),
                         "resid1": eng.resid[1].clone(), "h_mlp": eng.h_mlp.clone(), "s_wo": eng.s_wo.clone(), "s_down": eng.s_down.clone(),
                         "qkv": eng.qkv.clone(), "s_qkv": eng.s_qkv.clone(), "n_qkv": eng.n_qkv.value}
                    snaps[(i, stage)] = d
                eng(tok, pos, hook=hook)
                torch.cuda.synchronize()
                got[fast] = snaps
            L.teal_set_fast(1)
        for key in got[1]:
            a, b = got[1][key], got[0][key]
            line = []
            for nm in a:
                if nm in ("n_qkv",):
                    line.append(f"n_qkv {a[nm]}/{b[nm]}")
                    continue
                if nm in ("qkv", "s_qkv"):
                    continue
                x, y = a[nm].contiguous().view(-1), b[nm].contiguous().view(-1)
                xi = x.view(torch.int32) if x.dtype == torch.float32 else x.view(torch.int16)
                yi = y.view(torch.int32) if y.dtype == torch.float32 else y.view(torch.int16)
                nd = int((xi != yi).sum())
                if nd:
                    idx = torch.nonzero(xi != yi).view(-1)[:6].tolist()
                    line.append(f"{nm}: {nd} of {x.numel()} differ (first {idx}; lean {[float(x[j]) for j in idx[:3]]} general {[float(y[j]) for j in idx[:3]]})")
            print(f"layer {key[0]} after {key[1]}: " + ("; ".join(line) if line else "all equal"))
            if key[1] == "attn":
                hd = model.config.head_dim
                ns = eng.att_split
                x = a["att_ws"][: model.config.n_head * ns * (hd + 2)].view(model.config.n_head, ns, hd + 2)
                y = b["att_ws"][: model.config.n_head * ns * (hd + 2)].view(model.config.n_head, ns, hd + 2)
                ne = x.view(torch.int32) != y.view(torch.int32)
                for h in torch.nonzero(ne.any(2).any(1)).view(-1).tolist():
                    print(f"   head {h}: per split differing (m, l, o-count) {[(bool(ne[h, s_, 0]), bool(ne[h, s_, 1]), int(ne[h, s_, 2:].sum())) for s_ in range(ns)]}; "
                          f"m lean {x[h, :, 0].tolist()} general {y[h, :, 0].tolist()}; l lean {x[h, :, 1].tolist()} general {y[h, :, 1].tolist()}")
            if key[1] == "qkv":
                # lean q (rotated by the epilogue) vs the rotation of the general path's rounded slab sum, element by element
                n = b["n_qkv"]
                st = (n + 3) & ~3
                sl = b["s_qkv"].view(-1)[: eng.nqkv * st].view(eng.nqkv, st)
                acc = torch.zeros(eng.nqkv, dtype=torch.float32, device=DEV)
                for j in range(n):
                    acc = acc + sl[:, j]
                hd = model.config.head_dim
                qg = acc[: eng.dim].to(dt).float().view(-1, hd // 2, 2)
                cs = eng.rope.view(-1, hd // 2, 2)[NP].float()
                c_, s_ = cs[:, 0], cs[:, 1]
                pe = (qg[..., 1] * s_)
                po = (qg[..., 0] * s_)
                even = (qg[..., 0].double() * c_.double() - pe.double()).float()
                odd = (qg[..., 1].double() * c_.double() + po.double()).float()
                want = torch.stack((even, odd), dim=-1).reshape(-1).to(dt)
                ql = a["qkv"].view(-1)[: eng.dim]
                bad = torch.nonzero(want.view(torch.int16) != ql.view(torch.int16)).view(-1)
                print(f"   lean rotated q vs rotation of the general slabs' rounded sum: {bad.numel()} of {eng.dim} differ; heads {sorted(set((bad // hd).tolist()))}; "
                      f"first {[(int(j), float(ql[j]), float(want[j])) for j in bad[:4]]}")
            if key == (0, "qkv"):
                # the general path's q|k|v: slabs summed in slice order, rounded once; the lean path's v row must equal that v bit for bit
                n = b["n_qkv"]
                st = (n + 3) & ~3
                sl = b["s_qkv"].view(-1)[: eng.nqkv * st].view(eng.nqkv, st)
                acc = torch.zeros(eng.nqkv, dtype=torch.float32, device=DEV)
                for j in range(n):
                    acc = acc + sl[:, j]
                v_gen = acc[eng.dim + eng.nqkv - eng.dim - (eng.nqkv - eng.dim) // 2:].to(dt)
                kvs = (eng.nqkv - eng.dim) // 2
                v_gen = acc[eng.dim + kvs:].to(dt).view(-1)
                v_lean = a["v_row"].view(-1)
                print(f"   general slabs ({n}) -> v vs the lean epilogue's appended v row: {int((v_gen.view(torch.int16) != v_lean.view(torch.int16)).sum())} of {v_gen.numel()} differ")


main()
