"""CPU: host-side mirror of the reference interface (thresholds, greedy lookup, op schemas,
monkeypatch bundle, SparsifyFn) and the C-ABI library's export table. No GPU compute."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from helpers import GOLDEN, ROOT, load_kat

HIST = os.path.join(GOLDEN, "hist", "Llama-2-7B")
TAGS = (("attn_h1", "self_attn", "h1"), ("attn_h2", "self_attn", "h2"), ("mlp_h1", "mlp", "h1"), ("mlp_h2", "mlp", "h2"))


@pytest.fixture(scope="module")
def thresholds():
    with open(os.path.join(GOLDEN, "thresholds.json")) as f:
        return json.load(f)


def test_distribution_icdf_bit_identical_to_reference(thresholds):
    from teal_amd.distribution import Distribution, threshold_for_sparsity
    levels = thresholds["levels"]
    for layer in (0, 15):
        for tag, sub, h in TAGS:
            d = Distribution(os.path.join(HIST, f"layer-{layer}", sub), h)
            ref = thresholds["models"]["Llama-2-7B"][layer][tag]
            for s, want in zip(levels, ref):
                assert threshold_for_sparsity(d, s) == want, (layer, tag, s)
    # survey probe values: layer 0 / 15 at s = 0.5
    i5 = levels.index(0.5)
    l0, l15 = thresholds["models"]["Llama-2-7B"][0], thresholds["models"]["Llama-2-7B"][15]
    assert abs(l0["attn_h1"][i5] - 0.004945) < 1e-6 and abs(l15["mlp_h1"][i5] - 0.167918) < 1e-6


def test_icdf_numpy_crosscheck(thresholds, oracle):
    """independent numpy restatement of icdf agrees to fp32 rounding."""
    hist = torch.load(os.path.join(HIST, "layer-15", "mlp", "histograms.pt"), weights_only=True)
    ref = thresholds["models"]["Llama-2-7B"][15]["mlp_h1"]
    for s, want in zip(thresholds["levels"], ref):
        got = oracle.icdf_np(hist["h1"].numpy(), hist["h1_centers"].numpy(), 0.5 + 0.5 * s)
        assert abs(got - want) <= 2e-6 * max(1.0, abs(want))


def test_icdf_edges():
    from teal_amd.distribution import Distribution
    d = Distribution(os.path.join(HIST, "layer-0", "mlp"), "h2")
    assert d.icdf(0.0).item() == d.bin_centers[0].item()
    assert d.icdf(1.0 + 1e-3).item() == d.bin_centers[-1].item()
    qs = [0.5 + 0.05 * i for i in range(10)]
    vals = [d.icdf(q).item() for q in qs]
    assert all(b >= a for a, b in zip(vals, vals[1:]))
    assert abs(float(d.cdf(torch.tensor(vals[3]))) - qs[3]) < 5e-3


def test_histogram_file_format():
    h = torch.load(os.path.join(HIST, "layer-0", "mlp", "histograms.pt"), weights_only=True)
    for k in ("h1", "h1_centers", "h2", "h2_centers"):
        assert h[k].shape == (10000,) and h[k].dtype == torch.float32


def test_greedy_lookup_matches_reference(tmp_path):
    """fixture holds layers 0 and 31 of Llama-2-7B's lookup; present them as layer-0 / layer-1."""
    from teal_amd.utils import get_layer_greedy_sparsities
    with open(os.path.join(GOLDEN, "greedy_llama2_7b.json")) as f:
        g = json.load(f)
    for i, src in enumerate((0, 31)):
        os.makedirs(tmp_path / f"layer-{i}")
        os.symlink(os.path.join(GOLDEN, "lookup", "Llama-2-7B", f"layer-{src}", "results.csv"),
                   tmp_path / f"layer-{i}" / "results.csv")
    for t in ("0.3", "0.4", "0.5", "0.6"):
        got = get_layer_greedy_sparsities([float(t)] * 2, str(tmp_path))
        ref = g["targets"][t]["sparsities"]
        for p in got:
            assert got[p] == [ref[p][0], ref[p][31]], (t, p)
    # survey probe: target 0.5, layer 0
    r = g["targets"]["0.5"]["sparsities"]
    assert abs(r["q"][0] - 0.90) < 1e-9 and abs(r["down"][0] - 0.558) < 1e-3 and abs(r["gate"][31] - 0.409) < 1e-3


def test_greedy_thresholds_fixture_consistent():
    """F2's thresholds are icdf(0.5+0.5*s) of F2's sparsities on the raw histograms we hold."""
    from teal_amd.monkeypatch import layer_thresholds
    with open(os.path.join(GOLDEN, "greedy_llama2_7b.json")) as f:
        g = json.load(f)["targets"]["0.5"]
    got = layer_thresholds(0, HIST, g["sparsities"])
    for p, v in got.items():
        assert v == g["thresholds"][p][0], p
    got15 = layer_thresholds(15, HIST, g["sparsities"])
    assert all(got15[p] == g["thresholds"][p][15] for p in got15)
    assert len({got["q"], got["k"], got["v"]}) == 3  # greedy gives three distinct qkv thresholds


def test_sparsify_fn_matches_reference_rule():
    from teal_amd.utils import SparsifyFn

    class D:
        def icdf(self, q):
            return torch.tensor(0.25)

    k = load_kat("kat_boundary.npz")
    x = torch.from_numpy(k["x"].view(np.float16).copy()).view(1, 1, -1)
    fn = SparsifyFn(D())
    assert fn.get_threshold() == 0.0
    fn.threshold = float(k["sparsifyfn_tau"])
    out = fn(x).view(-1).numpy().view(np.uint16)
    assert np.array_equal(out, k["sparsifyfn_out"])
    fn.set_threshold(0.0)
    assert fn.threshold == 0.0
    fn.set_threshold(0.5)
    assert fn.threshold == 0.25 and fn.sparsity_level == 0.5
    # prefill: only the last half of the sequence is sparsified (utils/utils.py:36-43)
    xs = torch.full((1, 4, 8), 0.1, dtype=torch.float16)
    y = fn(xs)
    assert torch.equal(y[:, :2], xs[:, :2]) and (y[:, 2:] == 0).all()
    assert torch.equal(SparsifyFn(D(), init_threshold=0.25, apply_prefill=False)(xs), xs)


def test_op_schemas_match_reference():
    from teal_amd.kernels import SparseGEMV, SparseQKVGEMV
    assert SparseGEMV.schematize() == "(Tensor hidden_states, Tensor weights, float threshold, int sparsity_bin) -> Tensor"
    assert SparseQKVGEMV.schematize() == ("(Tensor x, Tensor weight, float threshold_q, float threshold_k, "
                                          "float threshold_v, int sparsity_bin, int kv_size) -> Tensor")


def test_schema_inference_general():
    from typing import List, Optional, Tuple
    from teal_amd.kernels.compile_wrapper import BaseKernel

    class K(BaseKernel):
        def forward(self, a: torch.Tensor, b: Optional[torch.Tensor], c: List[int], d: bool, e: torch.dtype) -> Tuple[torch.Tensor, torch.Tensor]:
            return a, a

    assert K.schematize() == "(Tensor a, Tensor? b, int[] c, bool d, ScalarType e) -> (Tensor, Tensor)"

    class Bad(BaseKernel):
        def forward(self, a, b: int) -> torch.Tensor:
            return a

    with pytest.raises(TypeError):
        Bad.schematize()


def test_ops_register_fake_and_refuse_cpu():
    from torch._subclasses.fake_tensor import FakeTensorMode
    from teal_amd.kernels import SparseGEMV, SparseQKVGEMV
    g = SparseGEMV.initialize("sparse_gemv", "cuda")
    op = g.operator(True)
    assert g.is_registered and op is torch.ops.teal.sparse_gemv
    assert SparseGEMV.initialize("sparse_gemv", "cuda").operator(True) is op  # idempotent
    q = SparseQKVGEMV.initialize("sparse_qkv_gemv", "cuda").operator(True)
    with FakeTensorMode():
        x = torch.empty(1, 1, 64, dtype=torch.bfloat16)
        w = torch.empty(192, 64, dtype=torch.bfloat16)
        y = op(x, w, 0.1, 0)
        assert y.shape == (1, 1, 192) and y.dtype == torch.bfloat16
        assert q(x, w, 0.1, 0.2, 0.3, 0, 64).shape == (1, 1, 192)
        assert op(torch.empty(1, 7, 64), torch.empty(192, 64), 0.1, 0).shape == (1, 7, 192)
    # no CPU implementation exists: the product path never falls back
    with pytest.raises(NotImplementedError):
        op(torch.zeros(1, 1, 64, dtype=torch.float16), torch.zeros(192, 64, dtype=torch.float16).T.contiguous().T, 0.1, 0)
    # un-compiled operator() is the python forward itself
    assert g.operator(False) == g.forward


def test_python_wrappers_keep_reference_asserts():
    from teal_amd.kernels import qkv_gemv, splitk_sparse_gemv
    x = torch.zeros(1, 1, 64, dtype=torch.float16)
    with pytest.raises(AssertionError):
        splitk_sparse_gemv(torch.zeros(1, 1, 32, dtype=torch.float16), torch.zeros(128, 64, dtype=torch.float16).T.contiguous().T, 0.1, 0)
    with pytest.raises(AssertionError, match="column major"):
        splitk_sparse_gemv(x, torch.zeros(128, 64, dtype=torch.float16), 0.1, 0)  # row-major weight
    with pytest.raises(AssertionError, match="column major"):
        qkv_gemv(x, torch.zeros(192, 64, dtype=torch.float16), 0.1, 0.1, 0.1, 0, 64)
    with pytest.raises(RuntimeError, match="GPU only"):
        splitk_sparse_gemv(x, torch.zeros(128, 64, dtype=torch.float16).T.contiguous().T, 0.1, 0)


def test_monkeypatch_layer_installs_reference_attribute_bundle():
    from teal_amd.gpt_fast.model import ModelArgs, TransformerBlock
    from teal_amd.monkeypatch import monkeypatch_layer
    cfg = ModelArgs(n_layer=1, n_head=4, n_local_heads=2, dim=64, intermediate_size=128, vocab_size=128, block_size=32)
    with torch.device("cpu"):
        layer = TransformerBlock(cfg).half()
    w1_before = layer.feed_forward.w1.weight.detach().clone()
    th = monkeypatch_layer(0, layer, 0.5, HIST, "cuda")
    ff, at = layer.feed_forward, layer.attention
    for name in ("gemv1_kernel", "gemv1", "gemv2_kernel", "gemv2", "thresh_up", "thresh_gate", "thresh_down", "sparsity_bin"):
        assert hasattr(ff, name), name
    for name in ("gemv1_kernel", "gemv1", "gemv2_kernel", "gemv2", "thresh_q", "thresh_k", "thresh_v", "thresh_o", "sparsity_bin"):
        assert hasattr(at, name), name
    assert ff.gemv1 is torch.ops.teal.sparse_gemv and at.gemv1 is torch.ops.teal.sparse_qkv_gemv
    assert ff.sparsity_bin == 0 and at.sparsity_bin == 0
    # uniform: q = k = v (all from attn_h1), gate = up (mlp_h1)  (generate.py:278-287)
    assert th["q"] == th["k"] == th["v"] == at.thresh_q and th["gate"] == th["up"] == ff.thresh_gate
    with open(os.path.join(GOLDEN, "thresholds.json")) as f:
        g = json.load(f)
    i5 = g["levels"].index(0.5)
    l0 = g["models"]["Llama-2-7B"][0]
    assert (at.thresh_q, at.thresh_o, ff.thresh_up, ff.thresh_down) == (l0["attn_h1"][i5], l0["attn_h2"][i5], l0["mlp_h1"][i5], l0["mlp_h2"][i5])
    # weights re-laid column-major, values unchanged
    from teal_amd.monkeypatch import ROW_PAD
    for lin in (ff.w1, ff.w3, ff.w2, at.wqkv, at.wo):
        N, Z = lin.weight.shape
        assert lin.weight.stride() == (1, N + ROW_PAD) and lin.weight.stride(1) > 1  # reference: stride(1) > 1
    assert torch.equal(ff.w1.weight, w1_before)
    # forwards swapped; old ones kept (model.py:222-224,270-272)
    assert hasattr(ff, "old_forward") and hasattr(at, "old_forward")


def test_c_abi_library_exports_every_declared_symbol():
    """include/teal_hip.h declares the product ABI and, under #ifdef TEAL_DIAGNOSTICS, the process-global tuning / phase-stamp
    switches.  libteal_hip.so (what the product path loads) exports every product symbol and NONE of the switches — no entry point
    changes how a later call behaves (SURVEY 8(b) "Ownership") — and libteal_hip_diag.so, the same sources with
    -DTEAL_DIAGNOSTICS, exports both sets."""
    import subprocess
    from teal_amd import _lib
    _lib.build()
    header = open(os.path.join(ROOT, "include", "teal_hip.h")).read()
    a, b = header.index("#ifdef TEAL_DIAGNOSTICS"), header.index("#endif /* TEAL_DIAGNOSTICS */")
    declared_diag = set(re.findall(r"\b(teal_[a-z_0-9]+)\s*\(", header[a:b]))
    declared = set(re.findall(r"\b(teal_[a-z_0-9]+)\s*\(", header[:a] + header[b:]))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert declared_diag == set(_lib.DIAG_EXPORTS), declared_diag ^ set(_lib.DIAG_EXPORTS)

    def dynamic_symbols(path):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        return {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("teal_")}

    assert dynamic_symbols(_lib.LIB_PATH) == declared                       # nothing undeclared, no switch, no "last launch" global
    assert dynamic_symbols(_lib.DIAG_LIB_PATH) == declared | declared_diag
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.teal_strerror.restype = ctypes.c_char_p
    L.teal_strerror.argtypes = [ctypes.c_int]
    assert L.teal_version() >= 100
    assert L.teal_strerror(0) == b"ok" and b"workspace" in L.teal_strerror(-5)
    L.teal_workspace_bytes.restype = ctypes.c_size_t
    assert L.teal_workspace_bytes(4096, 4096) >= 4096 * 4
    D = ctypes.CDLL(_lib.DIAG_LIB_PATH)
    assert D.teal_set_tuning(7, 0, 0, 0) == -8 and D.teal_set_tuning(0, 0, 0, 0) == 0
    # the loader declares signatures for all of them; load() is the product library unless the session asked for the other one
    lib = _lib.load()
    assert all(hasattr(lib, n) for n in _lib.EXPORTS)
    if os.environ.get("TEAL_LIB_FLAVOR", "") != "diag":
        assert not lib.teal_is_diagnostics_build and not any(hasattr(lib, n) for n in _lib.DIAG_EXPORTS)
    assert all(hasattr(_lib.load_diag(), n) for n in _lib.EXPORTS + _lib.DIAG_EXPORTS)


def test_no_packed_fp32_instruction_in_the_libraries(tmp_path):
    """Round 6: next to another process's skinny GEMM the low half of v_pk_fma_f32 results gets dropped on MI355X
    (profiles/r06_concurrent_packed_fp32.txt); the libraries are built without the packed-fp32 feature (teal_amd/_lib.py:
    NO_PACKED_FP32), and no kernel of either build may contain v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32.
    Nor v_fma_mixlo_f16 / v_fma_mixhi_f16: they round the exact a * b + c ONCE to fp16 where the oracle (and fp32 arithmetic
    followed by .to(fp16)) rounds twice (profiles/r06_fma_mixlo_rounding.txt); teal_common.h: float_to_bits keeps the
    conversion a separate v_cvt_f16_f32."""
    import shutil
    import subprocess
    from teal_amd import _lib
    _lib.build()
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    for lib in (_lib.LIB_PATH, _lib.DIAG_LIB_PATH):
        d = tmp_path / os.path.basename(lib)
        d.mkdir()
        shutil.copy(lib, d / "lib.so")
        subprocess.run([objdump, "--offloading", "lib.so"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(f for f in os.listdir(d) if f.endswith("gfx950"))
        assert len(objs) == len(_lib.SOURCES), objs   # one code object per translation unit
        n_inst = 0
        for f in objs:
            asm = subprocess.check_output([objdump, "-d", f], cwd=d, text=True)
            n_inst += asm.count("v_fma") + asm.count("v_fmac")
            hits = re.findall(r"v_pk_(?:fma|mul|add)_f32|v_pk_mov_b32|v_fma_mix(?:lo|hi)_f16", asm)
            assert not hits, (lib, f, hits[:3])
        assert n_inst > 1000   # (the disassembly was not empty)


def test_oracle_is_not_reachable_from_the_product_package():
    """teal_amd/ must never import, link or shell out to oracle/ (parity would be void)."""
    pkg = os.path.join(ROOT, "teal_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "teal_oracle" not in src and "from oracle" not in src and "import oracle" not in src, fn


@pytest.mark.parametrize("tag,dt", [("f16", torch.float16), ("bf16", torch.bfloat16)])
def test_int8_quantiser_module_match_reference_fixture(golden_dir, tag, dt):
    """teal_amd.quantize (host logic, CPU): same codes / scales as the reference quantiser and the same dense
    forward as its WeightOnlyInt8Linear on the captured inputs (tests/golden/kat_int8.npz, F8)."""
    import os
    from teal_amd.quantize import WeightOnlyInt8Linear, is_int8, quantize_model_int8, quantize_per_channel
    k = np.load(os.path.join(golden_dir, "kat_int8.npz"))
    w = torch.from_numpy(k[f"{tag}_w"].view(np.int16).copy()).view(dt)
    q, s = quantize_per_channel(w)
    assert np.array_equal(q.numpy(), k[f"{tag}_q"])
    assert np.array_equal(s.numpy().view(np.uint32), k[f"{tag}_scales_f32"].view(np.uint32))
    lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False, dtype=dt)
    lin.weight.data = w
    m = WeightOnlyInt8Linear.from_linear(lin)
    assert np.array_equal(m.scales.view(torch.int16).numpy().view(np.uint16), k[f"{tag}_scales"])
    x = torch.from_numpy(k[f"{tag}_x"].view(np.int16).copy()).view(dt).view(1, 1, -1)
    y = m(x).view(-1)
    want = torch.from_numpy(k[f"{tag}_y_dense"].view(np.int16).copy()).view(dt)
    assert torch.equal(y, want)  # same torch ops as the reference module on the same CPU
    # quantize_model_int8 swaps every Linear (projections and lm_head), nothing else
    box = torch.nn.Sequential(torch.nn.Linear(16, 8, bias=False, dtype=dt), torch.nn.LayerNorm(8), torch.nn.Linear(8, 24, bias=False, dtype=dt))
    quantize_model_int8(box)
    assert is_int8(box[0]) and is_int8(box[2]) and isinstance(box[1], torch.nn.LayerNorm)
    assert box[0].weight.dtype == torch.int8 and box[0].scales.dtype == dt


def test_to_column_major_int8_pads_128_bytes():
    from teal_amd.monkeypatch import to_column_major
    from teal_amd.quantize import WeightOnlyInt8Linear
    lin = torch.nn.Linear(64, 40, bias=False, dtype=torch.float16)
    m = WeightOnlyInt8Linear.from_linear(lin)
    before = m.weight.clone()
    to_column_major(m)
    assert m.weight.shape == (40, 64) and m.weight.stride() == (1, 40 + 128) and torch.equal(m.weight, before)
    to_column_major(lin)
    assert lin.weight.stride() == (1, 40 + 64)


def test_weight_images_keep_rows_on_128_byte_lines():
    """Every weight image the kernels gather from pads its rows by a multiple of 128 bytes at the real layer widths, so that a
    column tile's 128- / 256-byte row segment never straddles an extra memory line (the int4 image's first padding, 64 bytes,
    cost 7 % at 7B widths and 15 % at 70B)."""
    from teal_amd.monkeypatch import ROW_PAD, UP_SHIFT_BYTES
    from teal_amd.quantize import INT4_ROW_PAD_BYTES
    assert (ROW_PAD * 2) % 128 == 0 and UP_SHIFT_BYTES % 128 == 0 and INT4_ROW_PAD_BYTES % 128 == 0
    for N in (4096, 6144, 10240, 12288, 11008, 14336, 28672, 13824, 17920, 22016, 32000, 128256):
        assert ((N + ROW_PAD) * 2) % 128 == 0 and (N + 128) % 128 == 0 and (N + INT4_ROW_PAD_BYTES) % 128 == 0, N


def test_engine_grouped_query_split_choice_mirrors_the_launcher():
    """engine.py picks the split count of the grouped-query attention launch from the same LDS formula and limits as
    attention_split_impl (teal_attention.hip); a drift would silently fall back to the per-query-head kernel."""
    import re
    from teal_amd.gpt_fast.engine import DecodeEngine
    src = open(os.path.join(ROOT, "teal_amd", "csrc", "teal_attention.hip")).read()
    assert re.search(r"kGqaMinSeq\s*=\s*4096;", src) and re.search(r"kGqaMinSeq8\s*=\s*2048;", src)
    assert re.search(r"kGqaMaxLds\s*=\s*128\s*\*\s*1024;", src)
    eng_src = open(os.path.join(ROOT, "teal_amd", "gpt_fast", "engine.py")).read()
    assert "(rep == 8 and self.max_seq >= 2048) or (rep == 4 and self.max_seq >= 4096)" in eng_src and "> 128 * 1024" in eng_src
    f = DecodeEngine._gqa_lds_bytes
    # Llama-2-70B shapes: 8 query heads per KV head, head_dim 128, 16 k positions, 32 splits -> 512 rows per share
    assert f(8, 128, 16384, 32) == ((8 + 2) * 64 + 2 * 8 * 8 + 8 * max(512, 8 * 128)) * 4
    # a share's scores grow with the cache length and shrink with the split count
    assert f(8, 128, 131072, 32) > 128 * 1024 >= f(8, 128, 131072, 64)
    assert f(4, 64, 4096, 8) == ((4 + 2) * 32 + 2 * 4 * 8 + 4 * max(512, 8 * 64)) * 4


def test_measurement_scripts_compile():
    """scripts/ only run on the GPU box; a syntax error there would surface in the middle of a measurement session."""
    import glob
    import py_compile
    files = sorted(glob.glob(os.path.join(ROOT, "scripts", "*.py")) + glob.glob(os.path.join(ROOT, "scripts", "micro", "*.py")))
    assert len(files) >= 10
    for f in files + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        py_compile.compile(f, doraise=True)


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The Python side hands teal_gemv_in_t / teal_gemv_out_t / teal_prefill_in_t to the C ABI as ctypes structures: every field's
    offset and each structure's size must be what gcc lays out for include/teal_hip.h (a field added on one side only would shift
    every pointer behind it silently)."""
    import ctypes
    import subprocess
    from teal_amd.gpt_fast.engine import GemvIn, GemvOut
    from teal_amd.gpt_fast.prefill import PrefillIn
    structs = {"teal_gemv_in_t": GemvIn, "teal_gemv_out_t": GemvOut, "teal_prefill_in_t": PrefillIn}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "teal_hip.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        c, f, v = ln.split()
        got[(c, f)] = int(v)
    for cname, cls in structs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), (cname, got[(cname, "size")], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
