"""Shared helpers for the parity tests (test infrastructure: may import oracle/)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_kat(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: d[k] for k in d.files}


def kat_weights(O, k):
    """W^T bits of a KAT: stored for small cases, regenerated from the portable hash otherwise."""
    if "wT" in k:
        return k["wT"]
    return O.hash_uniform_c(int(k["Z"]) * int(k["N"]), int(k["seed_w"]), float(k["scale_w"]), int(k["dtype"]))


def ref_keys(k):
    return [n for n in k if n.startswith("y_ref_")]


def tolerance(O, truth, dtype):
    """|y - truth64| <= 1e-3*max(1,|truth|) + 1 ulp_out(truth)."""
    return 1e-3 * np.maximum(1.0, np.abs(truth)) + O.ulp16(truth, dtype)


def torch_from_bits(bits, dtype, device):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).copy())
    return t.view(torch.float16 if dtype == 0 else torch.bfloat16).to(device)


def bits_from_torch(t):
    import torch
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def colmajor_weight(wT_bits, Z, N, dtype, device):
    """torch weight of shape [N, Z] with strides (1, N) whose memory image is wT [Z][N]."""
    return torch_from_bits(wT_bits, dtype, device).view(Z, N).T


# ---- which build of the library a GPU test launches through ---------------------------------------------------------------
# libteal_hip.so (the product) has no switches; forcing the general kernel, a launch geometry or the workgroup-wide list, and
# reading teal_last_launch_desc, are features of libteal_hip_diag.so (same sources, -DTEAL_DIAGNOSTICS).  Tests run the
# production configuration through the PRODUCT build and everything else through the diagnostics build.
import contextlib  # noqa: E402
import functools  # noqa: E402


@contextlib.contextmanager
def lib_for(fast=1, wave_local=1, tuning=None, desc=False):
    """`with lib_for(fast) as L:` — the product library for the production configuration (lean kernel where the shape
    qualifies, automatic geometry, wave-local compaction, no launch description asked for); otherwise the diagnostics build
    with the switches set.  Everything CONSTRUCTED inside the block (engines, workspaces, op calls) launches through L."""
    from teal_amd import _lib
    if fast == 1 and wave_local == 1 and tuning is None and not desc:
        yield _lib.load()
        return
    with _lib.diagnostics() as D:
        D.teal_set_fast(int(fast))
        D.teal_set_wave_local(int(wave_local))
        if tuning is not None:
            assert D.teal_set_tuning(*tuning) == 0
        yield D


def with_diagnostics(fn):
    """decorator: the whole test runs through libteal_hip_diag.so (`_lib.load()` returns it inside); the switches are reset and
    the product library is back afterwards"""
    @functools.wraps(fn)
    def wrapper(*a, **k):
        from teal_amd import _lib
        with _lib.diagnostics():
            return fn(*a, **k)
    return wrapper
