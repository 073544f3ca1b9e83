"""teal_amd — MI355X-native (gfx950) implementation of TEAL's activation-sparsity decode hot path.

Layout:
  csrc/teal_kernels.hip   hand-written HIP kernels + the C ABI (include/teal_hip.h)
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
