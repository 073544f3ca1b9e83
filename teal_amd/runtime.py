"""Device-side plumbing shared by the ops: dtype codes, current stream, split-K workspace.

PyTorch owns every buffer; the C ABI only borrows raw pointers for a stream-ordered launch
(SURVEY §8(b) "Ownership").  The workspace (fp32 split-K slabs behind the library's header of arrival
counters and sampler scratch, teal_workspace_init) is one persistent tensor per (device, stream),
allocated and prepared outside graph capture (the warm-up call every capture needs), so captured
graphs see a static address and two streams never share a counter.
"""
from __future__ import annotations

import weakref

import torch

from . import _lib

F16, BF16 = 0, 1
_DTYPE_CODE = {torch.float16: F16, torch.bfloat16: BF16}
_workspaces: dict[tuple[int, int], torch.Tensor] = {}
_retired: list[torch.Tensor] = []  # capture workspaces that were replaced: graphs captured earlier still hold their pointers
_MAX_STREAM_WORKSPACES = 8           # per device; the least recently created per-stream entries beyond this are dropped
_inited: set[int] = set()


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dt]
    except KeyError:
        raise TypeError(f"teal_amd sparse GEMV supports float16/bfloat16 activations and weights, got {dt}") from None


def init() -> int:
    """Cache the current device's properties inside the library (must not first happen during capture)."""
    L = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError("teal_amd: no HIP device visible to PyTorch; the sparse GEMV path has no CPU fallback")
    cu = L.teal_init()
    if cu <= 0:
        _lib.check(cu, "teal_init")
    _inited.add(torch.cuda.current_device())
    return cu


def graph_capture(g: "torch.cuda.CUDAGraph", **kw):
    """`with graph_capture(g):` = torch.cuda.graph(g) with the stream-capture error mode a process with an RCCL process group
    needs.  torch captures in hipStreamCaptureModeGlobal by default: while ANY stream of the process is capturing, a
    capture-unsafe call from ANY thread fails.  ProcessGroupNCCL's watchdog thread polls the completion events of earlier
    collectives (hipEventQuery) every 100 ms; on this ROCm stack that query is refused under a global-mode capture
    ("operation not permitted when stream is capturing") and the watchdog aborts the whole process — measured with a one-rank
    group on the leased GPU (profiles/r06_nccl_one_rank_probe.txt: the default mode dies, thread-local mode captures and
    replays the all-reduce).  With a process group alive the capture therefore runs in thread-local mode: only the capturing
    thread's own calls are checked, which is all this package's captures rely on."""
    import torch.distributed as dist
    if "capture_error_mode" not in kw and dist.is_available() and dist.is_initialized():
        kw["capture_error_mode"] = "thread_local"
    return torch.cuda.graph(g, **kw)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _prepared(idx: int, nbytes: int) -> torch.Tensor:
    """A workspace tensor whose header the library has zeroed and registered (teal_workspace_init).  The registration
    is keyed by the device pointer, so it must end with the tensor: when the tensor is collected — and the caching
    allocator may hand the address to anything else — a finalizer releases it (teal_workspace_release).  Keep the
    tensor itself alive for as long as launches (or captured graphs) use its pointer; do not keep views instead."""
    L = _lib.load()
    with torch.cuda.device(idx):  # allocation, the header memset's stream and the library's device context: all device idx
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=torch.device("cuda", idx))
        _lib.check(L.teal_workspace_init(ws.data_ptr(), ws.numel() * 4, stream_ptr()), "teal_workspace_init")
    weakref.finalize(ws, L.teal_workspace_release, ws.data_ptr())
    return ws


def workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Persistent prepared workspace for ops launched on the current stream of `device`, grown on demand.

    Eager launches key on (device, stream), so two streams never share slabs or arrival counters.  A stream capture
    cannot allocate, and torch.cuda.graph captures on a stream of its own, so captures use the device's capture
    workspace, which is created (same size) whenever an eager one is: run one warm-up call before capturing, and
    replay graphs captured through the ops one at a time (a DecodeEngine owns its workspace instead)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _inited:
        with torch.cuda.device(idx):
            init()
    capturing = torch.cuda.is_current_stream_capturing()
    key = (idx, -1 if capturing else stream_ptr())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if capturing:
            raise RuntimeError("teal_amd: the workspace must be allocated before graph capture "
                               "(run one warm-up call, or teal_amd.runtime.reserve_workspace(Z, N))")
        for k in (key, (idx, -1)):
            old = _workspaces.get(k)
            if old is None or old.numel() * 4 < nbytes:
                if old is not None and k[1] == -1:
                    # a graph captured through the ops may hold this pointer (slabs AND the arrival-ticket header): never
                    # hand the memory back to the allocator while the process lives — park it
                    _retired.append(old)
                _workspaces.pop(k, None)
                _workspaces[k] = _prepared(idx, nbytes)  # (a replaced eager tensor's finalizer releases its registration)
        ws = _workspaces[key]
        # bound the per-stream entries of this device (each keeps ~ 8 x 2 x N x 4 bytes): oldest first, never the capture one
        mine = [k for k in _workspaces if k[0] == idx and k[1] != -1]
        drop = [k for k in mine[: max(0, len(mine) - _MAX_STREAM_WORKSPACES)] if k != key]
        if drop:
            # launches queued on the evicted entries' streams may still write slabs / tickets into them: let the device drain
            # before the allocator can hand the memory to anything else (eviction is rare: the ninth stream of a device)
            torch.cuda.synchronize(idx)
            for k in drop:
                del _workspaces[k]
    return ws


def drop_workspaces() -> None:
    """Forget the cached per-stream workspaces (their finalizers release the registrations once the device has drained):
    _lib.diagnostics() switches the library every later launch goes through, and a workspace is prepared per library."""
    if _workspaces or _retired:
        torch.cuda.synchronize()
    _workspaces.clear()


def reserve_workspace(Z: int, N: int, device=None) -> torch.Tensor:
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return workspace(device, int(_lib.load().teal_workspace_bytes(int(Z), int(N))))


def new_workspace(Z: int, N: int, device=None) -> torch.Tensor:
    """A prepared workspace of its own (not the per-stream one): for an object that launches / replays on whatever
    stream is current, one launch at a time, e.g. the decode engine."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _inited:
        with torch.cuda.device(idx):
            init()
    return _prepared(idx, int(_lib.load().teal_workspace_bytes(int(Z), int(N))))
