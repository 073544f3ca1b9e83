"""Loader for libteal_hip.so (the C ABI of include/teal_hip.h).

The product path has NO CPU fallback: if the HIP library cannot be built/loaded, every op
raises.  `build()` cross-compiles for gfx950 with hipcc (works without a GPU present).
"""
from __future__ import annotations

import concurrent.futures
import contextlib
import ctypes
import glob
import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
# translation units: host logic + GEMV ABI, attention + sampler, and the GEMV kernel instantiations split by
# (weight width, activation dtype) so that they compile in parallel
SOURCES = ("teal_kernels.hip", "teal_attention.hip", "teal_gemv_w16_f16.hip", "teal_gemv_w16_bf16.hip",
           "teal_gemv_w8_f16.hip", "teal_gemv_w8_bf16.hip", "teal_gemv_fast_f16.hip", "teal_gemv_fast_bf16.hip", "teal_gemv_int4.hip",
           "teal_gemv_fast_w8_f16.hip", "teal_gemv_fast_w8_bf16.hip", "teal_comparators.hip", "teal_prefill.hip")
# translation units whose kernels take their hot arguments as scalar parameters: the command processor preloads the
# first 11 dwords into SGPRs at wave launch (no scalar-cache miss before the first activation load)
PRELOAD = {"teal_gemv_fast_f16.hip": 12, "teal_gemv_fast_bf16.hip": 12, "teal_gemv_fast_w8_f16.hip": 12,
           "teal_gemv_fast_w8_bf16.hip": 12, "teal_attention.hip": 12}
# No packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any kernel of the library (round 6).  Measured on
# MI355X: while ANOTHER process's skinny rocBLAS / hipBLASLt GEMM (Tensile MT64x32x256 / MT32x16x256, a 24-token F.linear) shares
# the GPU, the LOW half of v_pk_fma_f32 results is dropped now and then for a whole row group — every decode step next to such a
# process differed (whole accumulators, even columns only, delta = -x_r * W[r, cols]); the same kernels with v_fma_mix_f32 /
# v_fmac_f32 in its place are bit-reproducible under the same load, and alone on the GPU both forms give the same bits
# (profiles/r06_concurrent_packed_fp32.txt).  That is an observation on THESE kernels: a minimal pure-HIP kernel with the same
# v_pk_fma_f32 form is not damaged under the same load (profiles/r06_pk_fma_reproducer.txt) and the cause is not established; the
# separation is empirical — every packed build of the product kernels fails every step, every non-packed build (all widths,
# dtypes, weight formats) passes.  With the feature off the compiler folds the fp16 -> fp32 conversion into v_fma_mix_f32
# (8 instructions per 16 bytes of weights instead of 8 conversions + 4 packed FMAs).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# Floating-point contraction follows the SOURCE, not the optimiser: `a += x * y` inside one expression is one fma in every
# instantiation (hipcc's default, -ffp-contract=fast, fuses wherever the DAG combiner sees fit — it saw fit differently in the
# ROPED and the slab-reading instantiation of decode_attention_split_kernel once the packed forms were gone, and the lean and the
# general decode step, specified to be bit-identical, were not: tests/test_soak.py).  The GEMV units go further and switch
# contraction off (explicit fmaf only: the note at the top of teal_gemv_kernel.h).
FP_CONTRACT = "-ffp-contract=on"
INCLUDE = os.path.join(_ROOT, "include")
OBJ_DIR = os.path.join(CSRC, "_obj")
LIB_PATH = os.path.join(_PKG, "libteal_hip.so")  # the in-tree PRODUCT library: what build() writes and what load() opens
# The DIAGNOSTICS build of the same sources (-DTEAL_DIAGNOSTICS): adds the process-global tuning / phase-stamp switches and
# teal_last_launch_desc (DIAG_EXPORTS).  The product path never loads it; benchmarks' sweeps, phase probes and the parity tests
# that force the general kernel or a launch geometry do, through load_diag() / diagnostics().  TEAL_LIB_FLAVOR=diag makes
# load() itself return it (a whole test session against the diagnostics build).
DIAG_LIB_PATH = os.path.join(_PKG, "libteal_hip_diag.so")
# TEAL_LIB_PATH: load() opens another build of the library (same C ABI) instead — same-box A/B of two builds.  build() never
# writes there (an override pointing at an older build must not be overwritten by the current tree), and symbols that build
# lacks are tolerated (OPTIONAL_WITH_OVERRIDE).
LIB_OVERRIDE = os.environ.get("TEAL_LIB_PATH") or None
OPTIONAL_WITH_OVERRIDE = ("teal_decode_attention_split_roped", "teal_prefill_gemm", "teal_prefill_resid_norm", "teal_prefill_attention")

# every symbol include/teal_hip.h declares
EXPORTS = (
    "teal_version", "teal_strerror", "teal_init", "teal_workspace_bytes", "teal_compact",
    "teal_sparse_gemv", "teal_sparse_qkv_gemv", "teal_dense_gemv", "teal_sparse_gateup_silu",
    "teal_get_config", "teal_fused_gemv", "teal_decode_attention", "teal_sample_topk", "teal_sparse_qkv_gemv_ld", "teal_decode_attention_masked", "teal_decode_attention_split", "teal_sparse_qkv_gemv_i8", "teal_decode_attention_split_slabs", "teal_sparse_qkv_gemv_i4",
    "teal_workspace_init", "teal_workspace_release", "teal_sample_topk_ws", "teal_decode_attention_split_ws", "teal_cmp_flag_gemv",
    "teal_decode_attention_split_roped",
    "teal_prefill_gemm", "teal_prefill_resid_norm", "teal_prefill_attention",
)

# what libteal_hip_diag.so exports on top (include/teal_hip.h, #ifdef TEAL_DIAGNOSTICS); libteal_hip.so must export NONE of them
DIAG_EXPORTS = ("teal_set_tuning", "teal_set_fast", "teal_set_wave_local", "teal_set_phase_buffer", "teal_set_phase_stride",
                "teal_last_launch_desc")

_lib = None    # what load() returns: the product library (or the diagnostics build under TEAL_LIB_FLAVOR=diag / diagnostics())
_diag = None   # the diagnostics build, once loaded


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libteal_hip.so (no CPU fallback exists)")


def _plan(lib_path: str, obj_dir: str, flags, force: bool, srcs, headers):
    """(compile jobs, object list) of one library build; a job is (cmd, obj)."""
    newest_header = max(os.path.getmtime(h) for h in headers)
    hipcc = _hipcc()
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_header):
            continue
        extra = []
        if os.path.basename(src) in PRELOAD:
            extra = ["-mllvm", f"-amdgpu-kernarg-preload-count={PRELOAD[os.path.basename(src)]}"]
        jobs.append(([hipcc, *flags, *extra, "-c", src, "-o", obj], obj))
    return jobs, objs


def build(force: bool = False, verbose: bool = False, out: str | None = None, extra_flags=(), diag: bool = True) -> str:
    """hipcc --offload-arch=gfx950 -> teal_amd/libteal_hip.so (the product library; in-tree, travels with the repo) and, with
    `diag`, teal_amd/libteal_hip_diag.so (the same sources with -DTEAL_DIAGNOSTICS: DIAG_EXPORTS on top).  One `hipcc -c` per
    translation unit and library, all in one parallel pool, then one link each.  Returns the product library's path.
    `out` + `extra_flags`: one more build next to them (e.g. an experiment build for a same-box A/B through TEAL_LIB_PATH);
    its objects live in their own directory and it is always rebuilt."""
    headers = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(INCLUDE, "teal_hip.h")]
    srcs = [os.path.join(CSRC, f) for f in SOURCES]
    base = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", f"-I{INCLUDE}", f"-I{CSRC}", *NO_PACKED_FP32, FP_CONTRACT]
    if out is not None:
        targets = [(out, os.path.join(OBJ_DIR, os.path.basename(out).replace(".", "_")), [*base, *extra_flags], True)]
    else:
        targets = [(LIB_PATH, OBJ_DIR, [*base, *extra_flags], force)]
        if diag:
            targets.append((DIAG_LIB_PATH, os.path.join(OBJ_DIR, "diag"), [*base, "-DTEAL_DIAGNOSTICS", *extra_flags], force))
    todo = []
    for lib_path, obj_dir, flags, f in targets:
        if not f and os.path.exists(lib_path) and all(os.path.getmtime(lib_path) >= os.path.getmtime(d) for d in srcs + headers):
            continue
        os.makedirs(obj_dir, exist_ok=True)
        jobs, objs = _plan(lib_path, obj_dir, flags, f, srcs, headers)
        todo.append((lib_path, jobs, objs))
    all_jobs = [j for _, jobs, _ in todo for j in jobs]

    def compile_one(job):
        if verbose:
            print(" ".join(job[0]))
        r = subprocess.run(job[0], stderr=subprocess.PIPE, text=True)
        # (the host pass of the same command line does not know the AMDGPU feature and says so once per function: not news)
        err = "\n".join(ln for ln in r.stderr.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in ln)
        if err.strip():
            print(err, file=sys.stderr)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, job[0])

    if all_jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(all_jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, all_jobs))
    for lib_path, _, objs in todo:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib_path + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(lib_path + ".tmp", lib_path)
    return out or LIB_PATH


def _open(path: str, diag: bool) -> ctypes.CDLL:
    """dlopen one build of the library and declare its signatures."""
    # PyTorch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch FIRST so that
    # our library binds to the HIP runtime torch's tensors and streams live in; loading ours first
    # would bring in a second runtime that sees no device context.
    import torch  # noqa: F401

    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). teal_amd has no CPU/eager fallback for the sparse GEMV path.")
    L = ctypes.CDLL(path)
    vp, ci, cf, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.teal_version.restype = ci
    L.teal_strerror.argtypes = [ci]
    L.teal_strerror.restype = ctypes.c_char_p
    L.teal_init.restype = ci
    L.teal_workspace_bytes.argtypes = [ci, ci]
    L.teal_workspace_bytes.restype = sz
    L.teal_compact.argtypes = [vp, cf, ci, ci, vp, vp, vp]
    L.teal_sparse_gemv.argtypes = [vp, vp, vp, cf, ci, ci, ci, vp, sz, vp]
    L.teal_sparse_qkv_gemv.argtypes = [vp, vp, vp, cf, cf, cf, ci, ci, ci, ci, ci, vp, sz, vp]
    L.teal_sparse_qkv_gemv_ld.argtypes = [vp, vp, ci, vp, cf, cf, cf, ci, ci, ci, ci, ci, vp, sz, vp]
    L.teal_sparse_qkv_gemv_i8.argtypes = [vp, vp, vp, vp, cf, cf, cf, ci, ci, ci, ci, ci, ci, vp, sz, vp]
    L.teal_dense_gemv.argtypes = [vp, vp, vp, ci, ci, ci, vp, sz, vp]
    L.teal_sparse_qkv_gemv_i4.argtypes = [vp, vp, vp, vp, cf, cf, cf, ci, ci, ci, ci, ci, ci, ci, vp, sz, vp]
    L.teal_sparse_gateup_silu.argtypes = [vp, vp, vp, vp, cf, cf, ci, ci, ci, vp, sz, vp]
    L.teal_fused_gemv.argtypes = [vp, vp, ci, ci, vp, sz, ctypes.POINTER(ci), vp]
    L.teal_sample_topk.argtypes = [vp, ci, ci, ci, cf, vp, vp, vp, vp, ci, vp]
    L.teal_sample_topk_ws.argtypes = [vp, ci, ci, ci, cf, vp, vp, vp, vp, ci, vp, sz, vp]
    L.teal_workspace_init.argtypes = [vp, sz, vp]
    L.teal_workspace_release.argtypes = [vp]
    L.teal_decode_attention_masked.argtypes = [vp, vp, vp, vp, vp, vp, vp, cf, ci, ci, ci, ci, ci, vp]
    L.teal_decode_attention_split.argtypes = [vp, vp, vp, vp, vp, vp, vp, cf, ci, ci, ci, ci, ci, vp, sz, ci, vp]
    L.teal_decode_attention_split_slabs.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, cf, ci, ci, ci, ci, ci, vp, sz, ci, vp]
    L.teal_decode_attention_split_ws.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, cf, ci, ci, ci, ci, ci, vp, sz, ci, vp, sz, vp]
    L.teal_cmp_flag_gemv.argtypes = [vp, vp, ci, vp, vp, cf, ci, ci, ci, vp]
    if hasattr(L, "teal_decode_attention_split_roped"):  # (absent from a pre-round-4 build loaded through TEAL_LIB_PATH)
        L.teal_decode_attention_split_roped.argtypes = [vp, vp, vp, vp, vp, vp, cf, ci, ci, ci, ci, ci, vp, sz, ci, vp, sz, vp]
    L.teal_decode_attention.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    L.teal_get_config.argtypes = [ci, ci, ci, ctypes.POINTER(ci)]
    if hasattr(L, "teal_prefill_gemm"):
        L.teal_prefill_gemm.argtypes = [vp, vp, ci, ci, vp, ci, ci, vp, sz, ci, ci, ci, ctypes.POINTER(ci), vp]  # (teal_prefill_in_t*, ...)
        L.teal_prefill_resid_norm.argtypes = [vp, vp, ci, vp, vp, ci, vp, cf, ci, vp, vp, vp, vp, ci, vp]
        L.teal_prefill_attention.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    for name in EXPORTS + (DIAG_EXPORTS if diag else ()):
        if LIB_OVERRIDE and name in OPTIONAL_WITH_OVERRIDE and not hasattr(L, name):
            continue  # an older build loaded for A/B: callers of this entry point fail with AttributeError when they reach it
        getattr(L, name)  # AttributeError if the .so is stale
        if getattr(L, name).restype is None:
            getattr(L, name).restype = ci
    if diag:
        L.teal_set_tuning.argtypes = [ci, ci, ci, ci]
        L.teal_set_phase_buffer.argtypes = [vp]
        L.teal_set_phase_stride.argtypes = [sz]
        L.teal_set_wave_local.argtypes = [ci]
        L.teal_set_fast.argtypes = [ci]
        L.teal_last_launch_desc.restype = ctypes.c_char_p
    L.teal_is_diagnostics_build = diag
    return L


def load() -> ctypes.CDLL:
    """The library every op of the package calls: libteal_hip.so, the product build (no tuning switches, no global launch
    state).  Raises if it is missing — never falls back.  TEAL_LIB_FLAVOR=diag: the diagnostics build instead (a whole test
    session against it); inside a `with diagnostics():` block, too."""
    global _lib
    if _lib is None:
        if os.environ.get("TEAL_LIB_FLAVOR", "") == "diag" and not LIB_OVERRIDE:
            _lib = load_diag()
        else:
            _lib = _open(LIB_OVERRIDE or LIB_PATH, False)
    return _lib


def load_diag() -> ctypes.CDLL:
    """libteal_hip_diag.so: the same sources with the process-global diagnostics switches (DIAG_EXPORTS).  Benchmarks, phase
    probes and A/B parity tests only; a workspace prepared through one library is plain memory to the other."""
    global _diag
    if _diag is None:
        _diag = _open(DIAG_LIB_PATH, True)
    return _diag


@contextlib.contextmanager
def diagnostics():
    """Inside the block `load()` returns the diagnostics build, so everything CONSTRUCTED in it (engines, workspaces, op
    calls) launches through libteal_hip_diag.so and obeys its switches; the switches are reset on the way out and the cached
    per-stream workspaces (prepared through one library, plain memory to the other) are dropped on both edges.  Tests and
    benchmarks only; not thread-safe — like the switches themselves."""
    global _lib
    from . import runtime
    prev = load()
    D = load_diag()
    runtime.drop_workspaces()
    _lib = D
    try:
        yield D
    finally:
        D.teal_set_tuning(0, 0, 0, 0)
        D.teal_set_fast(1)
        D.teal_set_wave_local(1)
        D.teal_set_phase_stride(0)
        D.teal_set_phase_buffer(None)
        runtime.drop_workspaces()
        _lib = prev


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().teal_strerror(rc).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")
