from .compile_wrapper import BaseKernel, NAMESPACE  # noqa: F401
from .sparse_gemv import (DenseGEMV, SparseGEMV, SparseQKVGEMV, compact, dense_gemv, qkv_gemv,  # noqa: F401
                          sparse_gateup_silu, splitk_sparse_gemv)
