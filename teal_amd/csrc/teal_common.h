// teal_common.h — types, launch parameters and device helpers shared by the translation units of
// libteal_hip.so (gfx950 / CDNA4, wave64).  Internal header: the public boundary is include/teal_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "teal_hip.h"

namespace teal {


typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kMaxSeg = 3;
constexpr int kMaxSplit = 32;


struct Seg {
    const void* w;  // weight image of this segment: row-major [Z][ld]
    void* y;        // output of this segment (element 0 = first column of the segment)
    float tau;
    int ld;      // row stride in elements
    int col0;    // first column of the segment inside a weight row
    int ncols;   // columns in the segment
    int tile0;   // first tile id of the segment
    int ws_off;  // column offset of the segment inside a workspace slab
    const void* scale;  // int8 weights: per-output-column scale (activation dtype), element 0 = column col0
};

// fused activation producers: what the workgroup computes before the mask
struct InSpec {
    int mode;                  // 0 plain x; 1 residual + slabs -> RMSNorm; 2 silu(gate) * up
    int nslabs;                // mode 1: fp32 slabs to fold into the residual
    const void* resid_in;      // mode 1: residual stream [Z] (or a table when row_index is set)
    const int* row_index;      // mode 1: optional device int: resid_in += row_index[0] * Z
    const float* slabs;        // mode 1: [nslabs][Z]
    const void* norm_w;        // mode 1: RMSNorm weight [Z]
    void* resid_out;           // mode 1: updated residual, written by workgroup 0
    const float* att;          // mode 4: attention partials [n_head][att_ns][head_dim + 2] = {max, sum, o[head_dim]}
    int att_hd;                // mode 4: head_dim (64 or 128)
    int att_ns;                // mode 4: partials per head (4 or 8)
    int slabs_il;              // mode 1: slabs are interleaved [Z][(nslabs + 3) & ~3] (one 16-byte load per element)
    const unsigned long long* masks;  // mode 3: keep masks (one per 64 activations) emitted by the producer
    float eps;
    int gate_act;              // mode 2: the gate half already holds round(silu(gate)): x = round(gate * up)
};

struct Params {
    InSpec in;
    const void* x;
    float* ws;  // [split][ws_ld] fp32 partial slabs
    int Z;
    int nseg;
    int ntiles;
    int split;
    int ws_ld;
    int cap;       // LDS list capacity (entries)
    int to_ws;     // 1: always write fp32 slabs (an epilogue kernel follows)
    int sl;        // wave-local + element-wise producer: the register cache holds only this slice's rounds
    int krt;       // wave-local: register-cache depth to launch (4, 8 or 16)
    int wl;        // 1: wave-local compaction (no cross-wave list, no barriers before the stream); cap = per-wave capacity
    int ws_il;     // 1: slabs written interleaved, ws[col * stride + slice], stride = (split + 3) & ~3
    int w8;        // 1: weights are int8 (per-column scales in seg[].scale), 8 columns = 8 bytes per lane
    int pair;      // 1: seg[0] = gate, seg[1] = up over the SAME column tile; epilogue silu(g)*u -> seg[0].y
    unsigned long long* mask_out;  // pair: keep masks of the output vs mask_tau for the next launch (or null)
    float mask_tau;
    unsigned long long* phase;  // optional: per-workgroup phase timestamps (teal_set_phase_buffer)
    unsigned* tickets;          // host side: arrival counters of the caller's prepared workspace (or null)
    // TEAL_OUT_QKV_ROPE (lean kernel, split == 1): RoPE + KV-cache append in the epilogue; rope == nullptr otherwise
    const uint16_t* rope; const int* rope_pos; uint16_t* kc; uint16_t* vc;
    int rope_hd, rope_max_seq;
    int act0;                   // rounded output of segment 0 goes through silu (and is rounded again)
    int sum32;                  // TEAL_OUT_SLAB_SUM: seg[0].y is fp32 and receives the unrounded slice-order sum (lean kernel only)
    Seg seg[kMaxSeg];
};


struct Config {
    int lpr, waves, split, unroll;
};

// Per-device immutable properties, cached the first time a device is used (teal_init() on that device; every entry
// point does it lazily, which must not first happen under stream capture).  Indexed by the HIP device ordinal.
struct DeviceCtx {
    int num_cu;       // 0: not initialised yet
    bool gqa_lds_ok;  // the grouped-query attention kernel may take kGqaMaxLds of LDS on this device
};
DeviceCtx* device_ctx();                   // context of the CURRENT device (nullptr: no device)
inline int num_cu_or(int dflt) {           // CU count of the current device, or dflt without a device (host-only queries)
    DeviceCtx* c = device_ctx();
    return c && c->num_cu > 0 ? c->num_cu : dflt;
}
bool attention_device_init();              // teal_attention.hip: per-kernel attributes of the current device
// teal_fused_gemv over int4 group-quantised weights (out->weight_bits == 4): teal_gemv_int4.hip
int fused_gemv_i4(const teal_gemv_in_t* in, const teal_gemv_out_t* out, int Z, int dtype, void* ws, size_t ws_bytes,
                  int* nslabs_out, hipStream_t st);

// Diagnostics / tuning switches.  The PRODUCT library (libteal_hip.so) has none: the values below are compile-time
// constants, no entry point can change how a launch is configured, and the only host-side state is the per-device
// properties and the workspace registry (SURVEY 8(b) "Ownership").  libteal_hip_diag.so (the same sources with
// -DTEAL_DIAGNOSTICS; benchmarks, phase stamps, lean-vs-general and forced-geometry parity tests) turns them into
// process-global variables behind teal_set_tuning / teal_set_fast / teal_set_wave_local / teal_set_phase_* and keeps
// teal_last_launch_desc — NOT thread-safe, never loaded by the product path.
#ifdef TEAL_DIAGNOSTICS
constexpr bool kDiagnostics = true;
extern Config g_override;
extern unsigned long long* g_phase;
extern size_t g_phase_stride;
extern int g_phase_seq;
extern int g_wave_local;
extern int g_fast;
extern char g_last_desc[160];  // template instantiation + grid of the most recent GEMV launch (teal_last_launch_desc)
// stamp buffer of a launch: GEMV launches stamp the start of the buffer or, in stride mode, the next region of it
inline unsigned long long* phase_next_region() {
    if (!g_phase) return nullptr;
    unsigned long long* p = g_phase + (size_t)g_phase_seq * g_phase_stride;
    if (g_phase_stride) ++g_phase_seq;
    return p;
}
inline unsigned long long* phase_start_only() { return g_phase_stride ? nullptr : g_phase; }              // single-launch probes
inline unsigned long long* phase_strided_only() { return (g_phase && g_phase_stride) ? phase_next_region() : nullptr; }
#else
constexpr bool kDiagnostics = false;
constexpr Config g_override = {0, 0, 0, 0};
constexpr int g_wave_local = 1;
constexpr int g_fast = 1;
inline unsigned long long* phase_next_region() { return nullptr; }
inline unsigned long long* phase_start_only() { return nullptr; }
inline unsigned long long* phase_strided_only() { return nullptr; }
#endif
// the description of the launch a call made (kernel instantiation + grid, as rocprofv3 prints it) goes to the CALLER's
// buffer (teal_gemv_out_t.desc); the diagnostics build also keeps the last one for teal_last_launch_desc
inline void publish_desc(const char* s, char* dst, int dst_bytes) {
    if (dst && dst_bytes > 0) snprintf(dst, (size_t)dst_bytes, "%s", s);
#ifdef TEAL_DIAGNOSTICS
    snprintf(g_last_desc, sizeof g_last_desc, "%s", s);
#endif
}

// ---- caller-owned workspace --------------------------------------------------------------------------------------
// A workspace prepared by teal_workspace_init() starts with a header the library owns (zeroed once by that call, re-armed
// by every launch that uses it): the per-tile arrival counters of the single-launch split-K GEMVs and the scratch of the
// multi-workgroup sampler.  The fp32 split-K slabs follow.  One workspace per stream, so two streams / graphs / devices
// can never share a counter; an unprepared workspace (plain memory) still works — split-K GEMVs then run as GEMV +
// ordered reduce launch and the sampler as a single workgroup.
constexpr int kSampCap = 512;      // candidates a chunk of 8192 logits may contribute
constexpr int kSampMaxGroups = 16;
constexpr size_t kSampSlotBytes = (size_t)kSampMaxGroups * kSampCap * 8 + 256;  // candidates, counts, ticket
constexpr int kTicketTiles = 4096;                                              // column tiles a ticketed launch may have
constexpr size_t kWsTicketBytes = (size_t)kTicketTiles * sizeof(unsigned);
constexpr size_t kWsSamplerOff = kWsTicketBytes;
constexpr size_t kWsHeaderBytes = (kWsTicketBytes + kSampSlotBytes + 255) & ~(size_t)255;
bool ws_prepared(const void* ws, size_t ws_bytes);  // registered by teal_workspace_init with at least the header
inline unsigned* ws_tickets(void* ws) { return reinterpret_cast<unsigned*>(ws); }
inline unsigned char* ws_sampler(void* ws) { return reinterpret_cast<unsigned char*>(ws) + kWsSamplerOff; }
inline float* ws_slabs(void* ws) { return reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ws) + kWsHeaderBytes); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// GEMV launchers, one translation unit per (weight width, activation dtype) so that the ~290 kernel
// instantiations compile in parallel: teal_gemv_{w16,w8}_{f16,bf16}.hip
hipError_t launch_gemv_w16_f16(const Params& p, size_t lds, const Config& c, hipStream_t st);
hipError_t launch_gemv_w16_bf16(const Params& p, size_t lds, const Config& c, hipStream_t st);
hipError_t launch_gemv_w8_f16(const Params& p, size_t lds, const Config& c, hipStream_t st);
hipError_t launch_gemv_w8_bf16(const Params& p, size_t lds, const Config& c, hipStream_t st);

__device__ __forceinline__ float bits_to_float(uint32_t b16, bool bf16) {
    if (bf16) return __uint_as_float(b16 << 16);
    _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)b16);
    return (float)h;
}

template <bool BF16>
__device__ __forceinline__ uint16_t float_to_bits(float f) {
    if (BF16) {
        // gfx950 converts in hardware (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN): one instruction where the
        // integer form (add 0x7FFF + lsb, shift, NaN test) took six — the RMSNorm producer rounds four times per element
        // (profiles/r04_layer_phase_timing_8b_bf16.txt: "x ready" 1.75 us against 1.05 us in fp16 before this)
        return __builtin_bit_cast(uint16_t, (__bf16)f);
    }
    // v_cvt_f16_f32 (round to nearest even) of a value whose producer the compiler cannot see (empty asm: no instruction).
    // Written as `(_Float16)f` the conversion of an fmaf() result is folded into v_fma_mixlo_f16 where the selector sees both —
    // and that instruction rounds the EXACT a * b + c once to fp16, where v_fma_f32 + v_cvt_f16_f32 (what the oracle and the
    // reference's fp32-then-.to(fp16) arithmetic do) round twice: 1741 of 2^26 random cases differ in the last place on MI355X
    // (scripts/micro/fma_mixlo_rounding_probe.hip, profiles/r06_fma_mixlo_rounding.txt).  Round 6: without packed fp32 the fold
    // happened in decode_attention_split_kernel's RoPE and not in the qkv projection's epilogue, and the lean and the general
    // decode step — specified to be bit-identical — differed in one head every few layers (tests/test_soak.py).
    asm("" : "+v"(f));
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}

// RoPE of one (even, odd) pair (gpt-fast/model.py:apply_rotary_emb: x0 * cos - x1 * sin, x1 * cos + x0 * sin), written with
// explicit fused multiply-adds so that every translation unit — whatever its -ffp-contract setting — rounds the same way:
// the qkv projection's epilogue (teal_gemv_fast.h, ROPE) and the attention launches must agree bit for bit.
__device__ __forceinline__ float rope_even(float x0, float x1, float c, float s) { return fmaf(x0, c, -(x1 * s)); }
__device__ __forceinline__ float rope_odd(float x0, float x1, float c, float s) { return fmaf(x1, c, x0 * s); }

// round(silu(round(sum))): the gate projection's output with the activation applied where it is computed (model.py:258)
template <bool BF16>
__device__ __forceinline__ uint16_t silu_bits(const float sum) {
    const float g16 = bits_to_float(float_to_bits<BF16>(sum), BF16);
    return float_to_bits<BF16>(g16 / (1.0f + expf(-g16)));
}

// keep rule of the reference kernel: float32(|x|) > float32(tau)  (kernels/sparse_gemv.py:75)
__device__ __forceinline__ bool keep_rule(float v, float tau) { return fabsf(v) > tau; }


// wave64 inclusive scan: 4 DPP row_shr steps inside each row of 16 lanes, then the three row totals
// are folded in through SGPRs (v_readlane) — no LDS traffic, unlike __shfl_up (ds_bpermute).
__device__ __forceinline__ int wave_incl_scan(int v, const int lane) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(v, 15);
    const int t1 = __builtin_amdgcn_readlane(v, 31);
    const int t2 = __builtin_amdgcn_readlane(v, 47);
    return v + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
}

// sum of a float over the 64 lanes of a wave (result valid in every lane)
__device__ __forceinline__ float wave_sum_f(float v) {
    auto shr = [](float a, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += shr(v, std::integral_constant<int, 0x111>{});
    v += shr(v, std::integral_constant<int, 0x112>{});
    v += shr(v, std::integral_constant<int, 0x114>{});
    v += shr(v, std::integral_constant<int, 0x118>{});
    const int iv = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 15)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 31)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 47)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 63));
}

// v[lane] + v[lane ^ OFF] for OFF = 8, 16, 32 — the butterfly steps of the row-group reductions — without the LDS
// crossbar round trip of __shfl_xor (ds_bpermute) where the ISA offers something cheaper: a DPP row rotate inside the
// 16-lane row (8), ds_swizzle's SWAP mode (16: no address VGPR, no LDS bank access); 32 stays a shuffle
// (v_permlane32_swap through the builtin returned both halves unswapped on this toolchain: not used).
// Same operands as the shuffle, a + b == b + a: bit-identical sums.
template <int OFF>
__device__ __forceinline__ float xor_add(const float v) {
    const int iv = __builtin_bit_cast(int, v);
    if constexpr (OFF == 8) {
        return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, 0x128, 0xf, 0xf, false));  // row_ror:8
    } else if constexpr (OFF == 16) {
        return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(iv, 0x401F));  // swizzle(SWAP, 16)
    } else {
        return v + __shfl_xor(v, OFF);
    }
}

// max of a float over the 64 lanes of a wave (result valid in every lane): DPP row shifts + four v_readlane, no LDS
// crossbar traffic (six __shfl_xor = ds_bpermute round trips otherwise)
__device__ __forceinline__ float wave_max_f(float v) {
    auto shr = [](float a, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, a), __builtin_bit_cast(int, a), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v = fmaxf(v, shr(v, std::integral_constant<int, 0x111>{}));
    v = fmaxf(v, shr(v, std::integral_constant<int, 0x112>{}));
    v = fmaxf(v, shr(v, std::integral_constant<int, 0x114>{}));
    v = fmaxf(v, shr(v, std::integral_constant<int, 0x118>{}));
    const int iv = __builtin_bit_cast(int, v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 15)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 31))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 47)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 63))));
}

// sum over the SL (16 or 8) consecutive lanes that hold the 16-byte slices of one K/V row
template <int SL>
__device__ __forceinline__ float row_slices_sum(float v) {
    if constexpr (SL == 16) {  // one DPP row: rotate-and-add, every lane ends with the total
        auto ror = [](float a, auto ctrl) {
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), decltype(ctrl)::value, 0xf, 0xf, false));
        };
        v += ror(v, std::integral_constant<int, 0x128>{});  // row_ror:8
        v += ror(v, std::integral_constant<int, 0x124>{});  // row_ror:4
        v += ror(v, std::integral_constant<int, 0x122>{});  // row_ror:2
        v += ror(v, std::integral_constant<int, 0x121>{});  // row_ror:1
        return v;
    } else {
#pragma unroll
        for (int d = 1; d < SL; d <<= 1) v += __shfl_xor(v, d);
        return v;
    }
}


__device__ __forceinline__ int lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}


}  // namespace teal
