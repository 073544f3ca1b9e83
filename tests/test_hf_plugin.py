"""HF-transformers plugin surface (A8): SparsifiedLinear + sparse_fns on a tiny random LlamaForCausalLM."""
import os

import pytest
import torch

transformers = pytest.importorskip("transformers")


def _tiny_llama(dtype=torch.float32):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).to(dtype).eval()


def _fake_histograms(path, n_layers=2, bins=400):
    """format-only calibration files (N(0,1) activations) via the producer of this repo"""
    from teal_amd.calibrate import find_histogram
    g = torch.Generator().manual_seed(1)
    for i in range(n_layers):
        for sub in ("self_attn", "mlp"):
            d = os.path.join(path, f"layer-{i}", sub)
            os.makedirs(d)
            h = {}
            for k in ("h1", "h2"):
                h[k], h[f"{k}_centers"] = find_histogram(torch.randn(20000, generator=g), bins)
            torch.save(h, os.path.join(d, "histograms.pt"))


def test_plugin_surface_and_semantics_cpu(tmp_path):
    from teal_amd.hf import SparsifiedLinear, sparsify_hf_model
    m = _tiny_llama()
    ids = torch.randint(0, 128, (1, 12), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        dense = m(ids).logits
    _fake_histograms(str(tmp_path))
    ctl = sparsify_hf_model(m, str(tmp_path))
    layer = m.model.layers[0]
    assert set(layer.mlp.sparse_fns.keys()) == {"gate", "up", "down"} and set(layer.self_attn.sparse_fns.keys()) == {"q", "k", "v", "o"}
    assert isinstance(layer.mlp.gate_proj, SparsifiedLinear) and isinstance(layer.self_attn.o_proj, SparsifiedLinear)
    ctl.reset_sparsities()  # threshold exactly 0.0 (utils/utils.py:28-30): nothing non-zero is dropped
    with torch.no_grad():
        assert torch.allclose(m(ids).logits, dense, atol=1e-6)
    ctl.set_uniform_sparsity(0.5)
    assert layer.mlp.sparse_fns["gate"].threshold > 0 and layer.self_attn.sparse_fns["o"].sparsity_level == 0.5
    with torch.no_grad():
        sparse = m(ids).logits
    assert not torch.allclose(sparse, dense, atol=1e-4)
    # the MLP wrapper computes exactly the reference's _mlp_forward (teal/mlp.py:49-55)
    x = torch.randn(1, 1, 64)
    mlp = layer.mlp
    fn = mlp.sparse_fns
    with torch.no_grad():
        want = mlp.down_proj.linear(fn["down"](mlp.act_fn(mlp.gate_proj.linear(fn["gate"](x))) * mlp.up_proj.linear(fn["up"](x))))
        assert torch.allclose(mlp(x), want, atol=1e-6)
    # prefill rule: only the last half of the sequence is sparsified; apply_prefill=False leaves it dense
    ctl.set_apply_prefill(False)
    with torch.no_grad():
        assert torch.allclose(m(ids).logits, dense, atol=1e-6)
    ctl.set_mlp_sparsity(0.3)
    ctl.set_self_attn_sparsity(0.2)
    assert layer.mlp.sparse_fns["up"].sparsity_level == 0.3 and layer.self_attn.sparse_fns["q"].sparsity_level == 0.2
    ctl.set_sparsities({"q": [0.1, 0.4], "down": [0.6, 0.7]})
    assert m.model.layers[1].self_attn.sparse_fns["q"].sparsity_level == 0.4 and m.model.layers[1].mlp.sparse_fns["down"].sparsity_level == 0.7


@pytest.mark.gpu
def test_sparsified_linear_decode_uses_hip_kernel(tmp_path):
    from teal_amd.hf import SparsifiedLinear
    from teal_amd.utils import SparsifyFn

    class D:
        def icdf(self, q):
            return torch.tensor(0.4)

    lin = torch.nn.Linear(512, 768, bias=False).half().cuda()
    fn = SparsifyFn(D())
    fn.set_threshold(0.5)
    sl = SparsifiedLinear(lin, fn)
    x = torch.randn(1, 1, 512, device="cuda", dtype=torch.float16)
    with torch.no_grad():
        y = sl(x)                                   # HIP sparse GEMV
        ref = torch.nn.functional.linear(fn(x), lin.weight)  # masked dense, reference semantics
        # the kernel path re-laid the linear's own weight out column-major in place (no second copy to go stale)
        assert lin.weight.stride(0) == 1 and lin.weight.shape == (768, 512) and y.shape == ref.shape
        assert torch.allclose(y.float(), ref.float(), atol=3e-3, rtol=3e-3)
        xb = torch.randn(1, 6, 512, device="cuda", dtype=torch.float16)
        assert torch.allclose(sl(xb), torch.nn.functional.linear(fn(xb), lin.weight))  # prefill: eager path
        # an in-place weight update and a wholesale replacement (load_state_dict(assign=True)) both reach the kernel path
        lin.weight.data.mul_(2.0)
        assert torch.allclose(sl(x).float(), 2 * y.float(), atol=6e-3, rtol=3e-3)
        lin.load_state_dict({"weight": torch.randn(768, 512, device="cuda", dtype=torch.float16) * 0.05}, assign=True)
        y3 = sl(x)
        assert torch.allclose(y3.float(), torch.nn.functional.linear(fn(x), lin.weight).float(), atol=3e-3, rtol=3e-3)
