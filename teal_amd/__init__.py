"""teal_amd — MI355X-native (gfx950) implementation of TEAL's activation-sparsity decode hot path.

Layout:
  csrc/                   hand-written HIP for gfx950 behind the C ABI of include/teal_hip.h:
                          teal_gemv_kernel.h (sparse GEMV template), teal_gemv_w{16,8}_{f16,bf16}.hip (instantiations),
                          teal_attention.hip (decode attention + sampler), teal_kernels.hip (host logic + GEMV ABI)
  quantize.py             int8 weight-only quantiser / module (feeds the int8 sparse GEMV)
  hf.py, calibrate.py     HF-transformers plugin surface; calibration producers
  _lib.py                 hipcc build + ctypes loader of libteal_hip.so (no CPU fallback)
  kernels/                reference-shaped operator boundary: splitk_sparse_gemv, qkv_gemv,
                          SparseGEMV / SparseQKVGEMV (torch.ops.teal.*)
  distribution.py         histograms.pt reader + icdf threshold math
  utils.py                SparsifyFn, greedy lookup reader
  monkeypatch.py          monkeypatch_layer(): installs gemv1/gemv2/thresh_* on gpt-fast-shaped layers
  gpt_fast/               decode harness (model + generate.py with --hist_path/--sparsity)
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401


def build(force: bool = False, verbose: bool = False) -> str:
    return _lib.build(force=force, verbose=verbose)
