"""Device-side plumbing shared by the ops: dtype codes, current stream, split-K workspace.

PyTorch owns every buffer; the C ABI only borrows raw pointers for a stream-ordered launch
(SURVEY §8(b) "Ownership").  The fp32 split-K workspace is one persistent tensor per device,
allocated outside graph capture (the warm-up call every capture needs), so captured graphs
see a static address.
"""
from __future__ import annotations

import torch

from . import _lib

F16, BF16 = 0, 1
_DTYPE_CODE = {torch.float16: F16, torch.bfloat16: BF16}
_workspaces: dict[int, torch.Tensor] = {}
_inited = False


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dt]
    except KeyError:
        raise TypeError(f"teal_amd sparse GEMV supports float16/bfloat16 activations and weights, got {dt}") from None


def init() -> int:
    """Cache device properties inside the library (must not first happen during capture)."""
    global _inited
    L = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError("teal_amd: no HIP device visible to PyTorch; the sparse GEMV path has no CPU fallback")
    cu = L.teal_init()
    if cu <= 0:
        _lib.check(cu, "teal_init")
    _inited = True
    return cu


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Persistent fp32 split-K scratch for `device`, grown on demand (never during capture)."""
    if not _inited:
        init()
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ws = _workspaces.get(idx)
    if ws is None or ws.numel() * 4 < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("teal_amd: split-K workspace must be allocated before graph capture "
                               "(run one warm-up call, or teal_amd.runtime.reserve_workspace(Z, N))")
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=torch.device("cuda", idx))
        _workspaces[idx] = ws
    return ws


def reserve_workspace(Z: int, N: int, device=None) -> torch.Tensor:
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return workspace(device, int(_lib.load().teal_workspace_bytes(int(Z), int(N))))
