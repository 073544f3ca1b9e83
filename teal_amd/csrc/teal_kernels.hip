// teal_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for TEAL's
// activation-sparsity decode hot path, behind the C ABI of include/teal_hip.h.
//
// Replaces (reference tree FasterDecoding/TEAL @ 2024-10-22):
//   kernels/sparse_gemv.py:50-83    splitk_sparse_gemv_kernel   -> sparse_gemv_kernel<>
//   kernels/sparse_gemv.py:152-194  qkv_kernel                  -> sparse_gemv_kernel<> (3 segments)
//   kernels/sparse_gemv.py:8-12     init_to_zero("Y") memset    -> gone (no accumulation into Y)
//   kernels/sparse_gemv.py:83       fp16 tl.atomic_add split-K  -> fp32 slabs + ordered reduce
//
// Design (DESIGN.md has the long form):
//   * One workgroup = one column tile (LPR lanes x 16 B = BN columns) x one even share of the
//     kept-row list.  Each workgroup re-derives the kept list itself: a wave64 ballot per 64
//     activations, a prefix sum over the ballot popcounts in LDS, then the (row, x) pairs of its
//     share are scattered into an LDS list in ascending row order.  Because shares are cut from the
//     compacted list (not from the raw Z range) every workgroup streams the same number of rows.
//   * Main loop: each wave walks the LDS list RPW = 64/LPR rows at a time; a lane issues U
//     independent 16-byte non-temporal loads (weights are read exactly once per token) before the
//     first FMA, fp32 accumulators, no LDS staging of weights (GEMV has no reuse).
//   * Reduction: shuffle across the RPW row groups of a wave, LDS across waves (fixed order),
//     fp32 slab per K-slice, second tiny kernel sums slabs in slice order and rounds once.
//     No atomics anywhere: bit-reproducible, and bf16 needs no special path.
//   * HBM-bound skinny GEMV: no MFMA on purpose (north_star).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "teal_hip.h"

namespace {

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
    int swizzle;   // 1: XOR-swizzle tiles inside aligned groups of 8 (XCD decorrelation)
    int sl;        // wave-local + element-wise producer: the register cache holds only this slice's rounds
    int krt;       // wave-local: register-cache depth to launch (4, 8 or 16)
    int wl;        // 1: wave-local compaction (no cross-wave list, no barriers before the stream); cap = per-wave capacity
    int ws_il;     // 1: slabs written interleaved, ws[col * stride + slice], stride = (split + 3) & ~3
    int w8;        // 1: weights are int8 (per-column scales in seg[].scale), 8 columns = 8 bytes per lane
    int pair;      // 1: seg[0] = gate, seg[1] = up over the SAME column tile; epilogue silu(g)*u -> seg[0].y
    unsigned long long* mask_out;  // pair: keep masks of the output vs mask_tau for the next launch (or null)
    float mask_tau;
    unsigned long long* phase;  // optional: per-workgroup phase timestamps (teal_set_phase_buffer)
    Seg seg[kMaxSeg];
};

__device__ __forceinline__ float bits_to_float(uint32_t b16, bool bf16) {
    if (bf16) return __uint_as_float(b16 << 16);
    _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)b16);
    return (float)h;
}

template <bool BF16>
__device__ __forceinline__ uint16_t float_to_bits(float f) {
    if (BF16) {
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
    _Float16 h = (_Float16)f;  // v_cvt_f16_f32, round-to-nearest-even
    return __builtin_bit_cast(uint16_t, h);
}

// keep rule of the reference kernel: float32(|x|) > float32(tau)  (kernels/sparse_gemv.py:75)
__device__ __forceinline__ bool keep_rule(float v, float tau) { return fabsf(v) > tau; }

template <bool BF16>
__device__ __forceinline__ void fma8(float (&acc)[8], const u32x4 w, const float xv) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t q = w[j];
        float lo, hi;
        if (BF16) {
            lo = __uint_as_float(q << 16);
            hi = __uint_as_float(q & 0xFFFF0000u);
        } else {
            const f16x2 h = __builtin_bit_cast(f16x2, q);
            lo = (float)h.x;
            hi = (float)h.y;
        }
        acc[2 * j] = fmaf(lo, xv, acc[2 * j]);
        acc[2 * j + 1] = fmaf(hi, xv, acc[2 * j + 1]);
    }
}

// int8 weights (weight-only quantisation, gpt-fast/quantize.py:339-357): 8 columns = 8 bytes per lane.
// v_cvt_f32_ubyte is a quarter-rate conversion and made the kernel VALU-bound; instead each byte is turned
// into an fp16 by v_perm_b32 alone: u = q ^ 0x80 (= q + 128, unsigned) under the exponent byte 0x64 is the
// half 0x64uu = 1024 + u exactly, and the mixed-precision FMA (v_fma_mix_f32) consumes halves at full rate.
// The constant 1024 + 128 = 1152 leaves once per column in the epilogue:
//     sum q*x = sum (1152 + q)*x - 1152 * sum x      (costs ~4 of fp32's 24 bits; outputs carry 8-11)
constexpr float kInt8Bias = 1152.0f;
template <bool BF16>
__device__ __forceinline__ void fma8(float (&acc)[8], const u32x2 w, const float xv) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t q = w[j] ^ 0x80808080u;
        const f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0x64646464u, q, 0x04010400u));  // bytes 0, 1
        const f16x2 hi = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0x64646464u, q, 0x04030402u));  // bytes 2, 3
        acc[4 * j] = fmaf((float)lo.x, xv, acc[4 * j]);
        acc[4 * j + 1] = fmaf((float)lo.y, xv, acc[4 * j + 1]);
        acc[4 * j + 2] = fmaf((float)hi.x, xv, acc[4 * j + 2]);
        acc[4 * j + 3] = fmaf((float)hi.y, xv, acc[4 * j + 3]);
    }
}

// ------------------------------------------------------------------------------------------------
// Workgroup-wide compaction of x against one threshold.
//   phase A: one ballot per 64-element chunk -> masks[] in LDS
//   phase B: wave 0 turns popcounts into an exclusive prefix (prefix[nch] = total kept)
// `nan_keeps`: GEMV mode — a NaN activation is kept so that it poisons the output like the
// reference's masked `0 * NaN` does; teal_compact uses the pure rule.
// ------------------------------------------------------------------------------------------------
template <int WAVES, bool BF16>
__device__ __forceinline__ void wg_ballot_prefix(const uint16_t* __restrict__ x, const int Z,
                                                 const float tau, const bool nan_keeps,
                                                 unsigned long long* masks, int* prefix) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nch = (Z + 63) >> 6;
    for (int c = wave; c < nch; c += WAVES) {
        const int m = (c << 6) + lane;
        bool k = false;
        if (m < Z) {
            const float v = bits_to_float(x[m], BF16);
            k = keep_rule(v, tau) || (nan_keeps && (v != v));
        }
        const unsigned long long mask = __ballot(k);
        if (lane == 0) masks[c] = mask;
    }
    __syncthreads();
    if (wave == 0) {
        int base = 0;
        for (int g0 = 0; g0 < nch; g0 += 64) {
            const int c = g0 + lane;
            const int v = (c < nch) ? __popcll(masks[c]) : 0;
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            if (c < nch) prefix[c] = base + incl - v;
            base += __shfl(incl, 63);
        }
        if (lane == 0) prefix[nch] = base;
    }
    __syncthreads();
}

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

// optional per-workgroup phase timestamps (constant 100 MHz clock, comparable across CUs)
__device__ __forceinline__ void stamp(const Params& p, int phase) {
    if (p.phase && threadIdx.x == 0) p.phase[(size_t)blockIdx.x * 8 + phase] = wall_clock64();
}

__device__ __forceinline__ int lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ------------------------------------------------------------------------------------------------
// The sparse GEMV.  grid = ntiles * split workgroups of WAVES*64 threads.
// ------------------------------------------------------------------------------------------------
template <int LPR, int WAVES, int U, bool BF16, int MODE, int KRT, bool PAIR, bool W8 = false>
__global__ __launch_bounds__(WAVES * 64) void sparse_gemv_kernel(const Params p) {
    constexpr int RPW = 64 / LPR;  // rows a wave touches per load instruction
    constexpr int BN = LPR * 8;    // columns per tile (8 per lane: 16 B of fp16/bf16, 8 B of int8)
    constexpr int WB = W8 ? 1 : 2;  // bytes per weight
    using wvec = typename std::conditional<W8, u32x2, u32x4>::type;
    constexpr int T = WAVES * 64;
    constexpr int STRIDE = WAVES * RPW;  // list entries consumed per workgroup step

    extern __shared__ __align__(16) unsigned char smem[];
    const int Z = p.Z;
    const int nch = (Z + 63) >> 6;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(smem);
    int* wavecnt = reinterpret_cast<int*>(masks + nch);
    float* sumsq = reinterpret_cast<float*>(wavecnt + 16);
    uint32_t* list = reinterpret_cast<uint32_t*>(sumsq + 16);
    float* red = reinterpret_cast<float*>(list + (p.wl ? (size_t)p.cap * WAVES : (size_t)p.cap));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Column tiles are interleaved over hardware blocks (block b runs on XCD b % 8, so every XCD
    // walks the whole row range).  An XCD-contiguous tile range was measured 10-25 % slower.
    int tile = blockIdx.x % p.ntiles;
    const int slice = blockIdx.x / p.ntiles;
    // Block b runs on XCD b % 8, so with tile = b % ntiles every XCD would only ever touch one
    // residue class (mod 8) of column tiles, i.e. of DRAM channels; XOR-ing the low 3 tile bits with
    // the next 3 keeps the set of tiles in flight identical but spreads each XCD over all residues.
    if (p.swizzle == 1 && tile < (p.ntiles & ~7)) tile = (tile & ~7) | ((tile ^ (tile >> 3)) & 7);
    if (p.swizzle >= 8 && tile < (p.ntiles & ~7)) tile = (tile & ~7) | ((tile + (p.swizzle - 8)) & 7);  // diagnostic rotation

    int s = 0;
    if (p.nseg > 1 && tile >= p.seg[1].tile0) s = 1;
    if (p.nseg > 2 && tile >= p.seg[2].tile0) s = 2;
    const Seg sg = p.seg[s];
    const int tcol0 = (tile - sg.tile0) * BN;  // first column of the tile inside the segment

    const uint16_t* __restrict__ x = reinterpret_cast<const uint16_t*>(p.x);
    stamp(p, 0);
    // int8: the per-column scales are needed only in the epilogue, where a dependent global load would add a full
    // (cold) memory round trip to every launch: thread t fetches the scale of tile column t right now
    uint32_t scb = 0u, scb2 = 0u;
    if constexpr (W8) {
        if (tid < BN && tcol0 + tid < sg.ncols) {
            scb = reinterpret_cast<const uint16_t*>(sg.scale)[tcol0 + tid];
            if constexpr (PAIR) scb2 = reinterpret_cast<const uint16_t*>(p.seg[1].scale)[tcol0 + tid];
        }
    }

    // ---- phase A: one ballot per 64 activations -> masks[]; the activations a wave ballots stay
    //      in its registers for the scatter (chunk c is owned by wave c % WAVES).  KRT (template) is
    //      the number of register-cached chunks per wave, sized to Z by the host, so every load below
    //      is unconditional (clamped address) and ALL of them are in flight before the first use. ----
    constexpr int PER = 64 / WAVES;  // owned chunks per group of 64 chunks
    constexpr int KR = KRT;
    constexpr int GREG = KR / PER;   // groups of 64 chunks covered by the register cache
    // PAIR: the list is the union of the two keep sets (smaller threshold); see the stream loop
    const float tau = PAIR ? fminf(p.seg[0].tau, p.seg[1].tau) : sg.tau;
    // register k of wave w caches chunk w + WAVES * k (round k of the wave).  Slice-local (wave-local compaction
    // with an element-wise producer): a workgroup only ever needs the rounds of ITS slice, so register k caches
    // round slice + k * split instead — 1/split of the loads, and vectors up to split * 16 rounds fit the cache
    const int kbase = p.sl ? slice : 0, kstep = p.sl ? p.split : 1;
    auto chunk_of = [&](const int k) { return wave + WAVES * (kbase + k * kstep); };
    uint32_t xr[KR];
    int mcl[KR];  // clamped element index of (k, lane)
#pragma unroll
    for (int k = 0; k < KR; ++k)
        mcl[k] = min((chunk_of(k) << 6) + lane, Z - 1);
    // activation of element m after the fused producer (modes 0 and 2 are element-wise)
    auto load_act = [&](const int m) -> uint32_t {
        if constexpr (MODE == 2) {
            // silu(gate) * up with the roundings of the unfused fp16/bf16 sequence (model.py:258-259)
            const float gt = bits_to_float(x[m], BF16);
            const float up = bits_to_float(x[Z + m], BF16);
            const float sl = bits_to_float(float_to_bits<BF16>(gt / (1.0f + expf(-gt))), BF16);
            return float_to_bits<BF16>(sl * up);
        } else {
            return (uint32_t)x[m];
        }
    };
    if constexpr (MODE == 1) {
        // h = resid + round(sum of split-K slabs);  x = round(round(h * rsqrt(mean(h^2) + eps)) * w)
        // (gpt-fast/model.py:158-161 residual adds, :289-291 RMSNorm) — every workgroup recomputes
        // it from L2-resident inputs; workgroup 0 stores the new residual stream.
        const uint16_t* resid = reinterpret_cast<const uint16_t*>(p.in.resid_in);
        if (p.in.row_index) resid += (size_t)p.in.row_index[0] * Z;
        const uint16_t* nw = reinterpret_cast<const uint16_t*>(p.in.norm_w);
        uint32_t rb[KR], wb[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            rb[k] = resid[mcl[k]];
            wb[k] = nw[mcl[k]];
        }
        float sacc[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) sacc[k] = 0.0f;
        if (p.in.slabs_il && p.in.nslabs > 0) {
            // producer wrote ws[col][slice]: all slabs of an element arrive in one (two) 16-byte loads,
            // issued together with the residual/weight loads above -> a single memory round trip
            const int stride = (p.in.nslabs + 3) & ~3;
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            f32x4 v0[KR], v1[KR];
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const f32x4* sp = reinterpret_cast<const f32x4*>(p.in.slabs + (size_t)mcl[k] * stride);
                v0[k] = sp[0];
                v1[k] = stride > 4 ? sp[1] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const int ns = p.in.nslabs;
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                float a = 0.0f;  // slab order 0,1,2,... (same order as the planar path and the reduce kernel)
#pragma unroll
                for (int j = 0; j < 4; ++j) a += (j < ns) ? v0[k][j] : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) a += (4 + j < ns) ? v1[k][j] : 0.0f;
                sacc[k] = a;
            }
        } else
        for (int q = 0; q < p.in.nslabs; q += 2) {  // two slabs per round trip, summed in slab order
            const bool two = q + 1 < p.in.nslabs;
            const float* s0 = p.in.slabs + (size_t)q * Z;
            const float* s1 = p.in.slabs + (size_t)(two ? q + 1 : q) * Z;
            float a0[KR], a1[KR];
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                a0[k] = s0[mcl[k]];
                a1[k] = s1[mcl[k]];
            }
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                sacc[k] += a0[k];
                sacc[k] += two ? a1[k] : 0.0f;
            }
        }
        float rv[KR];
        float ss = 0.0f;
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int m = (chunk_of(k) << 6) + lane;
            float r = bits_to_float(rb[k], BF16);
            if (p.in.nslabs > 0) {
                const float yv = bits_to_float(float_to_bits<BF16>(sacc[k]), BF16);
                r = bits_to_float(float_to_bits<BF16>(r + yv), BF16);
            }
            r = (m < Z) ? r : 0.0f;
            rv[k] = r;
            ss += r * r;
        }
        ss = wave_sum_f(ss);
        if (lane == 0) sumsq[wave] = ss;
        __syncthreads();
        float tot = (lane < WAVES) ? sumsq[lane] : 0.0f;
        tot = wave_sum_f(tot);
        const float rstd = rsqrtf(tot / (float)Z + p.in.eps);
        uint16_t* rout = reinterpret_cast<uint16_t*>(p.in.resid_out);
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int m = (chunk_of(k) << 6) + lane;
            const float xn = bits_to_float(float_to_bits<BF16>(rv[k] * rstd), BF16);
            xr[k] = (m < Z) ? (uint32_t)float_to_bits<BF16>(xn * bits_to_float(wb[k], BF16)) : 0u;
            if (rout && blockIdx.x == 0 && m < Z) rout[m] = float_to_bits<BF16>(rv[k]);
        }
    } else if constexpr (MODE == 4) {
        // x = attention output merged from 4 split-KV partials per head (flash-decoding): rescale by the
        // running maxima, sum, normalise, round once — the merge launch folded into the wo projection
        const int hd = p.in.att_hd, hs = hd + 2;
        // a wave's 64 consecutive elements lie in one head (head_dim 64 or 128, Z a multiple of it), so the
        // per-split {max, sum} are wave-uniform per chunk: lane j fetches them for (chunk j/NS, split j%NS) in
        // ONE load, turns them into the normalised weight e^(m - M) / L inside its group of NS lanes (DPP), and
        // the weights are broadcast with v_readlane — only the o[] values go through the vector memory pipe
        auto merge = [&](auto ns_tag) {
            constexpr int NS = decltype(ns_tag)::value;  // 4 or 8 partials per head
            constexpr int KM = (KR * NS <= 64) ? KR : 64 / NS;  // host refuses Z beyond KM chunks per wave
            const int kk = min(lane / NS, KM - 1), qq = lane % NS;
            const int mk = min(chunk_of(kk) << 6, Z - 1);
            const float2 st = *reinterpret_cast<const float2*>(p.in.att + ((size_t)(mk / hd) * NS + qq) * hs);
            float ov[KM][NS];
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const int h = mcl[k] / hd, d = mcl[k] - h * hd;
                const float* b = p.in.att + (size_t)h * NS * hs + 2 + d;
#pragma unroll
                for (int q = 0; q < NS; ++q) ov[k][q] = b[q * hs];
            }
#define TEAL_DPPF(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
            float M = fmaxf(st.x, TEAL_DPPF(st.x, 0xB1));  // quad_perm [1,0,3,2]
            M = fmaxf(M, TEAL_DPPF(M, 0x4E));              // quad_perm [2,3,0,1]
            if constexpr (NS == 8) M = fmaxf(M, TEAL_DPPF(M, 0x141));  // row_half_mirror: lane i <-> 7 - i
            const float f = st.y > 0.0f ? expf(st.x - M) : 0.0f;
            float Ls = st.y * f;
            Ls += TEAL_DPPF(Ls, 0xB1);
            Ls += TEAL_DPPF(Ls, 0x4E);
            if constexpr (NS == 8) Ls += TEAL_DPPF(Ls, 0x141);
#undef TEAL_DPPF
            const int cw = __float_as_int(f / Ls);
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                float Os = 0.0f;
#pragma unroll
                for (int q = 0; q < NS; ++q) Os += ov[k][q] * __int_as_float(__builtin_amdgcn_readlane(cw, NS * k + q));
                xr[k] = float_to_bits<BF16>(Os);
            }
#pragma unroll
            for (int k = KM; k < KR; ++k) xr[k] = 0u;
        };
        if (p.in.att_ns == 8) merge(std::integral_constant<int, 8>{});
        else merge(std::integral_constant<int, 4>{});
    } else if constexpr (MODE == 3) {
        // masks come from the producer (attention / gate|up epilogue): no compare, no ballot, and —
        // because nothing here depends on another wave — no barrier before the scatter either
#pragma unroll
        for (int k = 0; k < KR; ++k) xr[k] = x[mcl[k]];
    } else if constexpr (MODE == 2) {
        uint32_t gb[KR], ub[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) {  // all gate/up loads first, then the activation maths
            gb[k] = x[mcl[k]];
            ub[k] = x[Z + mcl[k]];
        }
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int m = (chunk_of(k) << 6) + lane;
            const float gt = bits_to_float(gb[k], BF16);
            const float sl = bits_to_float(float_to_bits<BF16>(gt / (1.0f + expf(-gt))), BF16);
            xr[k] = (m < Z) ? (uint32_t)float_to_bits<BF16>(sl * bits_to_float(ub[k], BF16)) : 0u;
        }
    } else {
#pragma unroll
        for (int k = 0; k < KR; ++k) xr[k] = x[mcl[k]];
    }
    const unsigned long long* gmask = MODE == 3 ? p.in.masks : masks;  // where chunk masks live
    int nloc = 0;                      // entries this wave/workgroup will stream
    const uint32_t* lp = list;         // where they are
    int estride = STRIDE;              // distance between the U entries a lane takes in one batch
    int eb = wave * RPW;               // first entry position of this wave
    if (p.wl) {
        // ---- wave-local compaction: every wave keeps the rows of the chunks it ballots itself (rounds
        //      k == slice mod split belong to this workgroup).  No cross-wave list, hence no scan and NO
        //      barrier between the activation and the first weight load.  Per-wave row counts differ by
        //      the binomial spread only; the launch is HBM-bound, so that does not cost time.
        uint32_t* mylist = list + (size_t)wave * p.cap;
        int base = 0, kmod = 0;
        unsigned long long mk[KR];
        if constexpr (MODE == 3) {
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int c = chunk_of(k);
                mk[k] = (c < nch) ? gmask[c] : 0ull;
            }
        }
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int c = chunk_of(k);
            const bool own = (p.sl || kmod == slice) && (c < nch);
            kmod = (kmod + 1 == p.split) ? 0 : kmod + 1;
            if (own) {
                unsigned long long mask;
                if constexpr (MODE == 3) {
                    mask = mk[k];
                } else {
                    const float v = bits_to_float(xr[k], BF16);
                    mask = __ballot(((c << 6) + lane < Z) && (keep_rule(v, tau) || (v != v)));
                }
                if ((mask >> lane) & 1ull) mylist[base + lane_rank(mask)] = ((uint32_t)((c << 6) + lane) << 16) | xr[k];
                base += __popcll(mask);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS is in-order per wave; keep the compiler honest
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        nloc = base;
        lp = mylist;
        estride = RPW;
        eb = 0;
        stamp(p, 1); stamp(p, 6); stamp(p, 2); stamp(p, 3);
    } else {
            if constexpr (MODE != 3) {
            int mycnt = 0;
        #pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int c = (k / PER) * 64 + wave + (k % PER) * WAVES;
                if (c < nch) {
                    const float v = bits_to_float(xr[k], BF16);
                    // NaN propagates like the reference's 0 * NaN on masked rows
                    const bool kp = ((c << 6) + lane < Z) && (keep_rule(v, tau) || (v != v));
                    const unsigned long long mask = __ballot(kp);
                    if (lane == 0) masks[c] = mask;
                    mycnt += __popcll(mask);
                }
            }
            for (int c = GREG * 64 + wave; c < nch; c += WAVES) {  // long vectors: beyond the register cache
                const int m = (c << 6) + lane;
                bool kp = false;
                if (m < Z) {
                    const float v = bits_to_float(load_act(m), BF16);
                    kp = keep_rule(v, tau) || (v != v);
                }
                const unsigned long long mask = __ballot(kp);
                if (lane == 0) masks[c] = mask;
                mycnt += __popcll(mask);
            }
            if (lane == 0) wavecnt[wave] = mycnt;
            stamp(p, 1);
            __syncthreads();
            stamp(p, 6);
        }

        // ---- phase B: every wave scans the chunk popcounts itself (DPP, no second barrier, no serial
        //      wave) and scatters the (row, x) pairs of its own chunks into the LDS list, ascending ------
        int total;
        if constexpr (MODE == 3) {
            int acc = 0;
            for (int c = lane; c < nch; c += 64) acc += __popcll(gmask[c]);
    #pragma unroll
            for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
            total = __builtin_amdgcn_readfirstlane(acc);
        } else {
            int t = (lane < WAVES) ? wavecnt[lane] : 0;
            t += __builtin_amdgcn_update_dpp(0, t, 0x111, 0xf, 0xf, false);
            t += __builtin_amdgcn_update_dpp(0, t, 0x112, 0xf, 0xf, false);
            t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xf, false);
            t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xf, false);
            total = __builtin_amdgcn_readlane(t, 15);  // WAVES <= 16: one DPP row holds every count
        }
        const int lo = (int)(((long long)total * slice) / p.split);
        const int hi = (int)(((long long)total * (slice + 1)) / p.split);
        nloc = hi - lo;
        stamp(p, 7);
        {
            int base = 0;
            auto scatter_group = [&](const int g0, const uint32_t* xg) {
                const int cg = g0 + lane;
                const int v = (cg < nch) ? __popcll(gmask[cg]) : 0;
                const int incl = wave_incl_scan(v, lane);
                const int excl = base + incl - v;
                base += __builtin_amdgcn_readlane(incl, 63);
    #pragma unroll
                for (int kk = 0; kk < PER; ++kk) {
                    const int j = wave + kk * WAVES;  // lane that holds an owned chunk's prefix (uniform)
                    const int c = g0 + j;
                    if (c >= nch) break;
                    const int pre = __builtin_amdgcn_readlane(excl, j);
                    const int cnt = __builtin_amdgcn_readlane(v, j);
                    if (pre + cnt <= lo || pre >= hi) continue;  // chunk outside this workgroup's share
                    const unsigned long long mask = gmask[c];
                    if ((mask >> lane) & 1ull) {
                        const int m = (c << 6) + lane;
                        const int pos = pre + lane_rank(mask);
                        const uint32_t xb = xg ? xg[kk] : load_act(m);
                        if (pos >= lo && pos < hi) list[pos - lo] = ((uint32_t)m << 16) | xb;
                    }
                }
            };
    #pragma unroll
            for (int g = 0; g < GREG; ++g)
                if (g * 64 < nch && base < hi) scatter_group(g * 64, &xr[g * PER]);
            for (int g0 = GREG * 64; g0 < nch && base < hi; g0 += 64) scatter_group(g0, nullptr);
        }
        stamp(p, 2);
        __syncthreads();
        stamp(p, 3);
    }

    // ---- stream the kept rows ----------------------------------------------------------------------
    const int g = lane / LPR;   // row group inside the wave
    const int cl = lane % LPR;  // 16-byte column slot inside the tile
    const int col = tcol0 + cl * 8;
    const bool col_ok = col < sg.ncols;  // ragged last tile
    const char* wp = reinterpret_cast<const char*>(sg.w) +
                     ((size_t)(sg.col0 + (col_ok ? col : 0))) * WB;
    const size_t ldb = (size_t)sg.ld * WB;
    // PAIR: the up-projection's tile (same columns) streamed with the same list
    const char* wp2 = PAIR ? reinterpret_cast<const char*>(p.seg[1].w) +
                                 ((size_t)(p.seg[1].col0 + (col_ok ? col : 0))) * WB : nullptr;
    const size_t ldb2 = PAIR ? (size_t)p.seg[1].ld * WB : 0;
    // PAIR with two different thresholds (block-wise greedy): the list holds the union (smaller tau);
    // a row is dropped from one of the two products by zeroing its weights (exactly a masked load)
    const float tau_g = p.seg[0].tau, tau_u = PAIR ? p.seg[1].tau : 0.0f;
    const bool two_tau = PAIR && (tau_g != tau_u);

    float acc[8], acc2[PAIR ? 8 : 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < (PAIR ? 8 : 1); ++j) acc2[j] = 0.0f;
    float xs = 0.0f, xs2 = 0.0f;  // W8: sum of the activations multiplied into acc / acc2 (bias correction)

    if (col_ok) {
        const int STEP = U * estride;
        auto full = [&](const int e) { return e + (U - 1) * estride + RPW <= nloc; };
        // issue the U (x2 for PAIR) 16-byte (int8: 8-byte) loads of one batch; nothing here waits
        auto issue = [&](wvec (&w)[U], wvec (&w2)[PAIR ? U : 1], float (&xv)[U], const int e0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t ent = lp[e0 + u * estride + g];
                xv[u] = bits_to_float(ent & 0xFFFFu, BF16);
                w[u] = __builtin_nontemporal_load(
                    reinterpret_cast<const wvec*>(wp + (size_t)(ent >> 16) * ldb));
                if constexpr (PAIR)
                    w2[u] = __builtin_nontemporal_load(
                        reinterpret_cast<const wvec*>(wp2 + (size_t)(ent >> 16) * ldb2));
            }
        };
        auto consume = [&](wvec (&w)[U], wvec (&w2)[PAIR ? U : 1], float (&xv)[U]) {
            if constexpr (W8) {
                // int8 weights are always finite: a row dropped from one of the two products is dropped by
                // zeroing its ACTIVATION for that product (which also keeps it out of the bias sum)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float xg = xv[u], xu = xv[u];
                    if (two_tau) {
                        const float ax = fabsf(xv[u]);
                        const bool nanx = xv[u] != xv[u];
                        if (!(ax > tau_g || nanx)) xg = 0.0f;
                        if (!(ax > tau_u || nanx)) xu = 0.0f;
                    }
                    fma8<BF16>(acc, w[u], xg);
                    xs += xg;
                    if constexpr (PAIR) {
                        fma8<BF16>(acc2, w2[u], xu);
                        xs2 += xu;
                    }
                }
            } else {
                if (two_tau) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float ax = fabsf(xv[u]);
                        const bool nanx = xv[u] != xv[u];
                        if (!(ax > tau_g || nanx)) w[u] = wvec(0u);
                        if constexpr (PAIR) if (!(ax > tau_u || nanx)) w2[u] = wvec(0u);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    fma8<BF16>(acc, w[u], xv[u]);
                    if constexpr (PAIR) fma8<BF16>(acc2, w2[u], xv[u]);
                }
            }
        };
        // two batches in flight per wave (software pipeline): the next batch's loads are issued before
        // the current batch is consumed, so a wave never sits with an empty memory queue
        wvec wa[U], wb[U], w2a[PAIR ? U : 1], w2b[PAIR ? U : 1];
        float xa[U], xb[U];
        bool fa = full(eb);
        if (fa) issue(wa, w2a, xa, eb);
        while (fa) {
            int ebn = eb + STEP;
            const bool fb = full(ebn);
            if (fb) issue(wb, w2b, xb, ebn);
            consume(wa, w2a, xa);
            eb = ebn;
            if (!fb) break;
            ebn = eb + STEP;
            fa = full(ebn);
            if (fa) issue(wa, w2a, xa, ebn);
            consume(wb, w2b, xb);
            eb = ebn;
        }
        // tail: clamp the entry index, zero the contribution of clamped lanes (fp16/bf16: zero weights;
        // int8: the zero activation alone does it, and it adds nothing to the bias sum)
        if (eb < nloc) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = eb + u * estride + g;
                const bool ok = e < nloc;
                const uint32_t ent = lp[ok ? e : nloc - 1];
                xa[u] = ok ? bits_to_float(ent & 0xFFFFu, BF16) : 0.0f;
                wvec t = __builtin_nontemporal_load(
                    reinterpret_cast<const wvec*>(wp + (size_t)(ent >> 16) * ldb));
                if (!W8 && !ok) t = wvec(0u);
                wa[u] = t;
                if constexpr (PAIR) {
                    wvec t2 = __builtin_nontemporal_load(
                        reinterpret_cast<const wvec*>(wp2 + (size_t)(ent >> 16) * ldb2));
                    if (!W8 && !ok) t2 = wvec(0u);
                    w2a[u] = t2;
                }
            }
            consume(wa, w2a, xa);
        }
    }

    stamp(p, 4);
    if (p.phase && lane == 0) p.phase[(size_t)gridDim.x * 8 + (size_t)blockIdx.x * 16 + wave] = wall_clock64();  // per-wave end of stream
    // ---- reduce: row groups of the wave, then waves (fixed order) --------------------------------
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], off);
        if constexpr (PAIR) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc2[j] += __shfl_xor(acc2[j], off);
        }
    }
    if (lane < LPR) {
        float* r = red + wave * BN + lane * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = acc[j];
        if constexpr (PAIR) {
            float* r2 = red + (WAVES + wave) * BN + lane * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) r2[j] = acc2[j];
        }
    }
    float* xsw = red + (PAIR ? 2 : 1) * WAVES * BN;  // W8: [2][WAVES] per-wave activation sums
    if constexpr (W8) {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            xs += __shfl_xor(xs, off);
            if constexpr (PAIR) xs2 += __shfl_xor(xs2, off);
        }
        if (lane == 0) {
            xsw[wave] = col_ok ? xs : 0.0f;
            if constexpr (PAIR) xsw[WAVES + wave] = col_ok ? xs2 : 0.0f;
        }
    }
    __syncthreads();
    // W8: sum q*x = sum (q + 1152)*x - 1152 * sum x (see fma8), then the per-column scale (quantize.py:354: the product is
    // scaled AFTER the reduction; here in fp32 before the single rounding)
    float bias = 0.0f, bias2 = 0.0f;
    if constexpr (W8) {
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) {
            bias += xsw[wv];
            if constexpr (PAIR) bias2 += xsw[WAVES + wv];
        }
        bias *= kInt8Bias;
        bias2 *= kInt8Bias;
    }
    if constexpr (PAIR) {
        // h = silu(gate) * up with the roundings of the unfused sequence (gpt-fast/model.py:258-259),
        // applied ONCE here instead of in every consumer workgroup; plus the keep masks of h against
        // the down-projection's threshold, so the consumer skips its compare/ballot phase entirely.
        static_assert(!PAIR || BN <= WAVES * 64, "one thread per tile column");
        if (tid < BN) {  // whole waves: BN is a multiple of 64
            const int c = tcol0 + tid;
            uint32_t hb = 0u;
            bool kp = false;
            if (c < sg.ncols) {
                float gs = 0.0f, us = 0.0f;
#pragma unroll
                for (int wv = 0; wv < WAVES; ++wv) {
                    gs += red[wv * BN + tid];
                    us += red[(WAVES + wv) * BN + tid];
                }
                if constexpr (W8) {
                    gs = (gs - bias) * bits_to_float(scb, BF16);
                    us = (us - bias2) * bits_to_float(scb2, BF16);
                }
                const float g16 = bits_to_float(float_to_bits<BF16>(gs), BF16);
                const float u16 = bits_to_float(float_to_bits<BF16>(us), BF16);
                const float sl = bits_to_float(float_to_bits<BF16>(g16 / (1.0f + expf(-g16))), BF16);
                hb = float_to_bits<BF16>(sl * u16);
                reinterpret_cast<uint16_t*>(sg.y)[c] = (uint16_t)hb;
                const float hv = bits_to_float(hb, BF16);
                kp = keep_rule(hv, p.mask_tau) || (hv != hv);
            }
            const unsigned long long mk = __ballot(kp);
            if (p.mask_out && lane == 0) p.mask_out[(tcol0 >> 6) + (tid >> 6)] = mk;
        }
    } else {
        static_assert(BN <= T, "one epilogue pass: thread t owns tile column t (prefetched int8 scale)");
        for (int t = tid; t < BN; t += T) {
            const int c = tcol0 + t;
            if (c >= sg.ncols) break;
            float sum = 0.0f;
#pragma unroll
            for (int wv = 0; wv < WAVES; ++wv) sum += red[wv * BN + t];
            if constexpr (W8) sum = (sum - bias) * bits_to_float(scb, BF16);  // t == tid: BN <= T, one pass
            if (p.split == 1 && !p.to_ws) {
                reinterpret_cast<uint16_t*>(sg.y)[c] = float_to_bits<BF16>(sum);
            } else if (p.ws_il) {
                p.ws[(size_t)(sg.ws_off + c) * ((p.split + 3) & ~3) + slice] = sum;
            } else {
                p.ws[(size_t)slice * p.ws_ld + sg.ws_off + c] = sum;
            }
        }
    }
    stamp(p, 5);
    if (p.phase && threadIdx.x == 0) {
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        p.phase[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)hwid << 32) | xcc;
    }
}

// y[n] = round(sum_s ws[s][n]) in slice order; one thread per column.
template <bool BF16>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const Params p) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= p.ws_ld) return;
    int s = 0;
    if (p.nseg > 1 && n >= p.seg[1].ws_off) s = 1;
    if (p.nseg > 2 && n >= p.seg[2].ws_off) s = 2;
    float sum = 0.0f;
    for (int k = 0; k < p.split; ++k) sum += p.ws[(size_t)k * p.ws_ld + n];
    reinterpret_cast<uint16_t*>(p.seg[s].y)[n - p.seg[s].ws_off] = float_to_bits<BF16>(sum);
}

// h[n] = silu(gate[n]) * up[n] from the fp32 slabs of a 2-segment (gate | up) GEMV.
// gate and up are rounded to dtype first, silu is rounded, then the product is rounded — the same
// roundings the unfused fp16 sequence F.silu(g) * u performs (gpt-fast/model.py:258-259).
template <bool BF16>
__global__ __launch_bounds__(256) void gateup_silu_epilogue_kernel(const Params p, void* h) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int N = p.seg[0].ncols;
    if (n >= N) return;
    float g = 0.0f, u = 0.0f;
    for (int k = 0; k < p.split; ++k) {
        g += p.ws[(size_t)k * p.ws_ld + n];
        u += p.ws[(size_t)k * p.ws_ld + N + n];
    }
    const float g16 = bits_to_float(float_to_bits<BF16>(g), BF16);
    const float u16 = bits_to_float(float_to_bits<BF16>(u), BF16);
    const float sl = g16 / (1.0f + expf(-g16));
    const float sl16 = bits_to_float(float_to_bits<BF16>(sl), BF16);
    reinterpret_cast<uint16_t*>(h)[n] = float_to_bits<BF16>(sl16 * u16);
}

// Standalone compaction (one workgroup): ascending kept indices + count to global memory.
template <bool BF16>
__global__ __launch_bounds__(1024) void compact_kernel(const uint16_t* __restrict__ x, const int Z,
                                                       const float tau, int32_t* __restrict__ idx_out,
                                                       int32_t* __restrict__ count_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nch = (Z + 63) >> 6;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(smem);
    int* prefix = reinterpret_cast<int*>(masks + nch);
    wg_ballot_prefix<16, BF16>(x, Z, tau, false, masks, prefix);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int c = wave; c < nch; c += 16) {
        const unsigned long long mask = masks[c];
        if ((mask >> lane) & 1ull) idx_out[prefix[c] + lane_rank(mask)] = (c << 6) + lane;
    }
    if (threadIdx.x == 0) *count_out = prefix[nch];
}

// ------------------------------------------------------------------------------------------------
// Single-token attention over a static KV cache (the step between gemv1 and gemv2 of
// gpt-fast/model.py:163-190): RoPE on q and the new k, KV-cache append, softmax(q K^T / sqrt(d)) V.
// One workgroup (256 threads) per query head; GQA by head group.  Rounding points follow the
// reference's fp16/bf16 tensors: rotated q/k, scores, probabilities and the output are rounded to
// dtype; accumulation is fp32.
// ------------------------------------------------------------------------------------------------
template <bool BF16, int NT, int HD>
__global__ __launch_bounds__(NT) void decode_attention_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ rope, const int* __restrict__ pos_ptr,
    uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache, uint16_t* __restrict__ y,
    unsigned long long* __restrict__ mask_out, const float mask_tau,
    const int n_head, const int n_kv, const int max_seq, const float scale, unsigned long long* __restrict__ phase) {
    constexpr int NW = NT / 64;
    constexpr int hd = HD;
    auto stamp_a = [&](const int i) { if (phase && threadIdx.x == 0) phase[(size_t)blockIdx.x * 8 + i] = wall_clock64(); };
    stamp_a(0);
    constexpr int SL = HD / 8;   // 16-byte slices per row (16 for hd=128, 8 for hd=64)
    constexpr int RW = 64 / SL;  // V rows per wave step (4 or 8)
    constexpr int VPF = 256 / (NW * RW) > 0 ? 256 / (NW * RW) : 1;  // V steps prefetched: the first 256 rows (hd=128)
    extern __shared__ __align__(16) unsigned char smem[];
    float* qs = reinterpret_cast<float*>(smem);  // [hd] rotated q
    float* kn = qs + hd;                         // [hd] rotated new k
    float* vn = kn + hd;                         // [hd] new v
    float* red = vn + hd;                        // [2 * NW] block reductions
    float* part = red + 2 * NW;                  // [NW][hd] per-wave partial outputs
    float* sc = part + NW * hd;                  // [max_seq] scores / probabilities
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x;
    const int rep = n_head / n_kv;
    const int kvh = h / rep;
    const int pos = pos_ptr[0];
    const int dim = n_head * hd, kvs = n_kv * hd;
    const uint16_t* qh = qkv + (size_t)h * hd;
    const uint16_t* kh = qkv + dim + (size_t)kvh * hd;
    const uint16_t* vh = qkv + dim + kvs + (size_t)kvh * hd;
    uint16_t* kc = k_cache + (size_t)kvh * max_seq * hd;
    uint16_t* vc = v_cache + (size_t)kvh * max_seq * hd;

    // ---- everything that only depends on `pos` is requested first: this thread's cached K row and
    //      its V slices are in flight while q/k are rotated (one memory round trip instead of three)
    // (lane = (row-in-wave rw, 16-byte slice ds): a wave reads 64/SL whole rows = 1 KiB contiguous per load)
    const int ds = lane % SL, rw = lane / SL;
    u32x4 kreg[VPF], vreg[VPF];
#pragma unroll
    for (int i = 0; i < VPF; ++i) {
        const int t = wave * RW + rw + i * NW * RW;
        kreg[i] = *reinterpret_cast<const u32x4*>(kc + (size_t)(t < pos ? t : 0) * hd + ds * 8);
    }
#pragma unroll
    for (int i = 0; i < VPF; ++i) {
        const int t = wave * RW + rw + i * NW * RW;
        vreg[i] = *reinterpret_cast<const u32x4*>(vc + (size_t)(t < pos ? t : 0) * hd + ds * 8);
    }

    // RoPE on interleaved pairs (model.py apply_rotary_emb), table rows are (cos, sin) in dtype
    if (tid < hd / 2) {
        const float c = bits_to_float(rope[((size_t)pos * (hd / 2) + tid) * 2], BF16);
        const float sn = bits_to_float(rope[((size_t)pos * (hd / 2) + tid) * 2 + 1], BF16);
        const float q0 = bits_to_float(qh[2 * tid], BF16), q1 = bits_to_float(qh[2 * tid + 1], BF16);
        const float k0 = bits_to_float(kh[2 * tid], BF16), k1 = bits_to_float(kh[2 * tid + 1], BF16);
        const uint16_t qa = float_to_bits<BF16>(q0 * c - q1 * sn), qb = float_to_bits<BF16>(q1 * c + q0 * sn);
        const uint16_t ka = float_to_bits<BF16>(k0 * c - k1 * sn), kb = float_to_bits<BF16>(k1 * c + k0 * sn);
        qs[2 * tid] = bits_to_float(qa, BF16);
        qs[2 * tid + 1] = bits_to_float(qb, BF16);
        kn[2 * tid] = bits_to_float(ka, BF16);
        kn[2 * tid + 1] = bits_to_float(kb, BF16);
        if (h % rep == 0) {  // one writer per KV head
            kc[(size_t)pos * hd + 2 * tid] = ka;
            kc[(size_t)pos * hd + 2 * tid + 1] = kb;
        }
    } else if (tid >= 128 && tid < 128 + hd) {
        const int d = tid - 128;
        const uint16_t vb = vh[d];
        vn[d] = bits_to_float(vb, BF16);
        if (h % rep == 0) vc[(size_t)pos * hd + d] = vb;
    }
    __syncthreads();
    stamp_a(1);

    // scores: each lane multiplies its 8-dim slice, the SL lanes of a row are summed with DPP; the new
    // token's own key comes from LDS, not from the cache line being written
    float qv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = qs[ds * 8 + j];
    float lmax = -INFINITY;
    auto score_row = [&](const int t, const u32x4 w) {
        float a = 0.0f;
        if (t == pos) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a += qv[j] * kn[ds * 8 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a += qv[2 * j] * bits_to_float(w[j] & 0xFFFFu, BF16);
                a += qv[2 * j + 1] * bits_to_float(w[j] >> 16, BF16);
            }
        }
        a = row_slices_sum<SL>(a);
        const float sv = bits_to_float(float_to_bits<BF16>(a * scale), BF16);
        if (t <= pos) {
            if (ds == 0) sc[t] = sv;
            lmax = fmaxf(lmax, sv);
        }
    };
#pragma unroll
    for (int i = 0; i < VPF; ++i) score_row(wave * RW + rw + i * NW * RW, kreg[i]);
    for (int tb = VPF * NW * RW; tb <= pos; tb += NW * RW) {  // beyond the prefetched rows (wave-uniform trip count)
        const int t = tb + wave * RW + rw;
        const u32x4 w = *reinterpret_cast<const u32x4*>(kc + (size_t)(t < pos ? t : 0) * hd + ds * 8);
        score_row(t, w);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, d));
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    stamp_a(2);
    float mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float lsum = 0.0f;
    for (int t = tid; t <= pos; t += NT) {
        const float e = expf(sc[t] - mx);
        sc[t] = e;
        lsum += e;
    }
    lsum = wave_sum_f(lsum);
    if (lane == 0) red[NW + wave] = lsum;
    __syncthreads();
    stamp_a(3);
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[NW + w];
    const float inv = 1.0f / tot;

    // output: 16-byte slices of V rows; lane = (row-in-wave rw, 8-dim slice ds)
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.0f;
    auto accum = [&](const int t, const u32x4 w) {
        const float pr = bits_to_float(float_to_bits<BF16>(sc[t] * inv), BF16);
        if (t == pos) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += pr * vn[ds * 8 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[2 * j] += pr * bits_to_float(w[j] & 0xFFFFu, BF16);
                o[2 * j + 1] += pr * bits_to_float(w[j] >> 16, BF16);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < VPF; ++i) {
        const int t = wave * RW + rw + i * NW * RW;
        if (t <= pos) accum(t, vreg[i]);
    }
#pragma unroll 4
    for (int t = wave * RW + rw + VPF * NW * RW; t <= pos; t += NW * RW) {
        const u32x4 w = (t < pos) ? *reinterpret_cast<const u32x4*>(vc + (size_t)t * hd + ds * 8) : (u32x4){0u, 0u, 0u, 0u};
        accum(t, w);
    }
    for (int off = SL; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += __shfl_xor(o[j], off);
    }
    if (lane < SL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wave * hd + lane * 8 + j] = o[j];
    }
    __syncthreads();
    stamp_a(4);
    if (tid < hd) {  // whole waves (hd = 64 or 128)
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += part[w * hd + tid];
        const uint16_t yb = float_to_bits<BF16>(acc);
        y[(size_t)h * hd + tid] = yb;
        if (mask_out) {  // keep masks of y for the wo projection (TEAL_IN_MASKED consumer)
            const float yv = bits_to_float(yb, BF16);
            const unsigned long long mk = __ballot(keep_rule(yv, mask_tau) || (yv != yv));
            if (lane == 0) mask_out[((size_t)h * hd + tid) >> 6] = mk;
        }
    }
    stamp_a(5);
}

// ------------------------------------------------------------------------------------------------
// Long contexts: split the cached positions of a head over `nsplit` workgroups (flash-decoding).
// Each workgroup produces an un-normalised partial {running max m, sum l, o[hd]} over its range; the
// merge kernel rescales and sums them, rounds once and emits the keep masks for the wo projection.
// With one workgroup per head a 4k context would leave 224 CUs idle while 32 stream 2 MB each.
// ------------------------------------------------------------------------------------------------
template <bool BF16, int HD, int NT>
__global__ __launch_bounds__(NT) void decode_attention_split_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ rope, const int* __restrict__ pos_ptr,
    uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache, float* __restrict__ partials,
    const int n_head, const int n_kv, const int max_seq, const int nsplit, const int chunk_max, const float scale,
    const float* __restrict__ qkv_slabs, const int qkv_nslabs) {
    constexpr int NW = NT / 64, hd = HD, SL = HD / 8, RW = 64 / SL;
    extern __shared__ __align__(16) unsigned char smem[];
    float* qs = reinterpret_cast<float*>(smem);
    float* kn = qs + hd;
    float* vn = kn + hd;
    float* red = vn + hd;            // [2 * NW]
    float* part = red + 2 * NW;      // [NW][hd]
    float* sc = part + NW * hd;      // [chunk_max]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
    const int rep = n_head / n_kv, kvh = h / rep;
    const int pos = pos_ptr[0], n = pos + 1;
    const int chunk = (n + nsplit - 1) / nsplit;
    const int t0 = sp * chunk, t1 = min(n, t0 + chunk);
    float* out = partials + (size_t)blockIdx.x * (hd + 2);
    if (t0 >= t1) {  // empty range (short sequence, many splits)
        if (tid < hd) out[2 + tid] = 0.0f;
        if (tid == 0) { out[0] = -INFINITY; out[1] = 0.0f; }
        return;
    }
    const bool has_new = (t1 == n);  // this workgroup's range ends with the token being decoded
    const int dim = n_head * hd, kvs = n_kv * hd;
    const uint16_t* qh = qkv + (size_t)h * hd;
    const uint16_t* kh = qkv + dim + (size_t)kvh * hd;
    const uint16_t* vh = qkv + dim + kvs + (size_t)kvh * hd;
    uint16_t* kc = k_cache + (size_t)kvh * max_seq * hd;
    uint16_t* vc = v_cache + (size_t)kvh * max_seq * hd;
    // lanes = (row rw, 16-byte slice ds): a wave load covers RW whole cache rows (coalesced).  The cached
    // rows depend only on pos, not on this step's q: the first PF row groups of K AND V are requested
    // before anything else, so their latency hides behind the q/rope loads, the rope and both barriers.
    const int ds = lane % SL, rw = lane / SL;
    constexpr int PF = 4, STEP = NW * RW;
    const int trow = t0 + wave * RW + rw;
    u32x4 kreg[PF], vreg[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int t = trow + i * STEP;
        const size_t off = (size_t)((t < t1 && t != pos) ? t : t0) * hd + ds * 8;
        kreg[i] = *reinterpret_cast<const u32x4*>(kc + off);
        vreg[i] = *reinterpret_cast<const u32x4*>(vc + off);
    }
    // element `col` of the qkv projection: the rounded vector, or (qkv_slabs) the fp32 split-K slabs of the
    // projection launch, interleaved [col][(nslabs + 3) & ~3], summed in slice order and rounded once here —
    // a narrow (GQA) wqkv can then be row-sliced over all CUs without a reduce launch in between
    auto qkv_at = [&](const uint16_t* base, const int i) -> uint16_t {
        if (!qkv_slabs) return base[i];
        const float* sp = qkv_slabs + (size_t)((base - qkv) + i) * ((qkv_nslabs + 3) & ~3);
        float a = 0.0f;
        for (int q = 0; q < qkv_nslabs; ++q) a += sp[q];
        return float_to_bits<BF16>(a);
    };
    if (tid < hd / 2) {
        const float c = bits_to_float(rope[((size_t)pos * (hd / 2) + tid) * 2], BF16);
        const float sn = bits_to_float(rope[((size_t)pos * (hd / 2) + tid) * 2 + 1], BF16);
        const float q0 = bits_to_float(qkv_at(qh, 2 * tid), BF16), q1 = bits_to_float(qkv_at(qh, 2 * tid + 1), BF16);
        qs[2 * tid] = bits_to_float(float_to_bits<BF16>(q0 * c - q1 * sn), BF16);
        qs[2 * tid + 1] = bits_to_float(float_to_bits<BF16>(q1 * c + q0 * sn), BF16);
        if (has_new) {
            const float k0 = bits_to_float(qkv_at(kh, 2 * tid), BF16), k1 = bits_to_float(qkv_at(kh, 2 * tid + 1), BF16);
            const uint16_t ka = float_to_bits<BF16>(k0 * c - k1 * sn), kb = float_to_bits<BF16>(k1 * c + k0 * sn);
            kn[2 * tid] = bits_to_float(ka, BF16);
            kn[2 * tid + 1] = bits_to_float(kb, BF16);
            if (h % rep == 0) {
                kc[(size_t)pos * hd + 2 * tid] = ka;
                kc[(size_t)pos * hd + 2 * tid + 1] = kb;
            }
        }
    } else if (has_new && tid >= 128 && tid < 128 + hd) {
        const int d = tid - 128;
        const uint16_t vb = qkv_at(vh, d);
        vn[d] = bits_to_float(vb, BF16);
        if (h % rep == 0) vc[(size_t)pos * hd + d] = vb;
    }
    __syncthreads();
    float qv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = qs[ds * 8 + j];
    float lmax = -INFINITY;
    auto score_row = [&](const int t, const u32x4 w) {
        float a = 0.0f;
        if (t == pos) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a += qv[j] * kn[ds * 8 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a += qv[2 * j] * bits_to_float(w[j] & 0xFFFFu, BF16);
                a += qv[2 * j + 1] * bits_to_float(w[j] >> 16, BF16);
            }
        }
        a = row_slices_sum<SL>(a);
        const float sv = bits_to_float(float_to_bits<BF16>(a * scale), BF16);
        if (t < t1) {
            if (ds == 0) sc[t - t0] = sv;
            lmax = fmaxf(lmax, sv);
        }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (t0 + i * STEP < t1) score_row(trow + i * STEP, kreg[i]);  // workgroup-uniform guard
#pragma unroll 4
    for (int tb = t0 + PF * STEP; tb < t1; tb += STEP) {
        const int t = tb + wave * RW + rw;
        score_row(t, *reinterpret_cast<const u32x4*>(kc + (size_t)((t < t1 && t != pos) ? t : t0) * hd + ds * 8));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, d));
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    float mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float lsum = 0.0f;
    for (int t = t0 + tid; t < t1; t += NT) {
        const float e = expf(sc[t - t0] - mx);
        sc[t - t0] = e;
        lsum += e;
    }
    lsum = wave_sum_f(lsum);
    if (lane == 0) red[NW + wave] = lsum;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[NW + w];
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.0f;
    auto pv_row = [&](const int t, const u32x4 w) {
        const float pr = sc[t - t0];
        if (t == pos) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += pr * vn[ds * 8 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[2 * j] += pr * bits_to_float(w[j] & 0xFFFFu, BF16);
                o[2 * j + 1] += pr * bits_to_float(w[j] >> 16, BF16);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (trow + i * STEP < t1) pv_row(trow + i * STEP, vreg[i]);
#pragma unroll 4
    for (int t = trow + PF * STEP; t < t1; t += STEP) pv_row(t, *reinterpret_cast<const u32x4*>(vc + (size_t)t * hd + ds * 8));
    for (int off = SL; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += __shfl_xor(o[j], off);
    }
    if (lane < SL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wave * hd + lane * 8 + j] = o[j];
    }
    __syncthreads();
    if (tid < hd) {
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += part[w * hd + tid];
        out[2 + tid] = acc;
    }
    if (tid == 0) { out[0] = mx; out[1] = tot; }
}

template <bool BF16>
__global__ __launch_bounds__(128) void decode_attention_merge_kernel(const float* __restrict__ partials,
                                                                     uint16_t* __restrict__ y,
                                                                     unsigned long long* __restrict__ mask_out,
                                                                     const float mask_tau, const int hd, const int nsplit) {
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float* p = partials + (size_t)h * nsplit * (hd + 2);
    if (tid >= hd) return;  // hd = 64 or 128: whole waves
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, p[(size_t)s * (hd + 2)]);
    float L = 0.0f, O = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
        const float* ps = p + (size_t)s * (hd + 2);
        if (ps[1] > 0.0f) {
            const float f = expf(ps[0] - M);
            L += ps[1] * f;
            O += ps[2 + tid] * f;
        }
    }
    const uint16_t yb = float_to_bits<BF16>(O / L);
    y[(size_t)h * hd + tid] = yb;
    if (mask_out) {
        const float yv = bits_to_float(yb, BF16);
        const unsigned long long mk = __ballot(keep_rule(yv, mask_tau) || (yv != yv));
        if (lane == 0) mask_out[((size_t)h * hd + tid) >> 6] = mk;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused sampler (gpt-fast/generate.py:49-66): logits / T -> keep the top-k -> softmax -> exponential-
// race multinomial (argmax p_i / q_i, q_i ~ Exp(1)), no host sync.  One workgroup; the k-th largest
// logit is found EXACTLY by a two-pass radix select on the 16-bit keys (ties at the pivot are all
// kept, as `logits < pivot -> -inf` does).  Randomness: counter-based hash of (seed, draw counter,
// index); the draw counter lives on the device and is bumped by the kernel, so hipGraph replays
// draw fresh numbers.  Token streams are not pinned by the reference (they depend on torch's RNG).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t order_key16(uint32_t b, bool bf16) {
    (void)bf16;  // fp16 and bf16 share sign-magnitude ordering
    return (b & 0x8000u) ? (~b & 0xFFFFu) : (b | 0x8000u);
}

__device__ __forceinline__ uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// sel[0] = the bin b (counted from the top) in which the `need`-th largest key falls, sel[1] = its rank
// inside that bin.  hist[256] -> suffix counts by a Hillis-Steele scan (all threads must call this).
__device__ __forceinline__ void select_bin(const unsigned int* hist, unsigned int* suf, unsigned int* sel,
                                           const unsigned int need, const int tid) {
    if (tid < 256) suf[tid] = hist[tid];
    __syncthreads();
#pragma unroll
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned int v = (tid < 256 && tid + d < 256) ? suf[tid + d] : 0u;
        __syncthreads();
        if (tid < 256) suf[tid] += v;
        __syncthreads();
    }
    if (tid < 256) {
        const unsigned int above = tid < 255 ? suf[tid + 1] : 0u;  // keys in strictly higher bins
        if (suf[tid] >= need && above < need) { sel[0] = (unsigned int)tid; sel[1] = need - above; }
    }
    if (tid == 0 && suf[0] < need) { sel[0] = 0u; sel[1] = need; }  // fewer keys than requested: keep all
    __syncthreads();
}

template <bool BF16>
__global__ __launch_bounds__(1024) void sample_topk_kernel(const uint16_t* __restrict__ logits, const int V,
                                                            const int top_k, const float inv_temp,
                                                            unsigned long long* __restrict__ rng_state,
                                                            int* __restrict__ token_out, int* __restrict__ pos_inout,
                                                            int* __restrict__ history, const int history_len) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int whist[16][256];  // per-wave sub-histograms: logits cluster in a few bins, a single
                                             // shared histogram serialises on LDS atomics
    __shared__ float fred[16];
    __shared__ int ired[16];
    __shared__ unsigned int sel[2];
    __shared__ unsigned int suf[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool filter = top_k > 0 && top_k < V;
    const int V8 = V >> 3;  // 16-byte vectors (vocab sizes are multiples of 8; the tail is handled scalar)
    const u32x4* lv = reinterpret_cast<const u32x4*>(logits);
    uint32_t pivot_key = 0;  // keep keys >= pivot_key
    float mx = -INFINITY;
    for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
    __syncthreads();
    // pass 1: high-byte histogram of the order-preserving 16-bit keys + global max
    for (int i = tid; i < V8; i += 1024) {
        const u32x4 w = lv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t lo = w[j] & 0xFFFFu, hi = w[j] >> 16;
            mx = fmaxf(mx, fmaxf(bits_to_float(lo, BF16), bits_to_float(hi, BF16)));
            if (filter) {
                atomicAdd(&whist[wave][order_key16(lo, BF16) >> 8], 1u);
                atomicAdd(&whist[wave][order_key16(hi, BF16) >> 8], 1u);
            }
        }
    }
    for (int i = (V8 << 3) + tid; i < V; i += 1024) {
        mx = fmaxf(mx, bits_to_float(logits[i], BF16));
        if (filter) atomicAdd(&whist[wave][order_key16(logits[i], BF16) >> 8], 1u);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (lane == 0) fred[wave] = mx;
    __syncthreads();
    if (tid < 256) {
        unsigned int a = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) a += whist[w][tid];
        hist[tid] = a;
    }
    __syncthreads();
    mx = fred[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, fred[w]);
    if (filter) {
        // suffix counts over the 256 bins (parallel scan), then the bin holding the top_k-th key
        select_bin(hist, suf, sel, (unsigned int)top_k, tid);
        __syncthreads();
        const unsigned int hb = sel[0], need2 = sel[1];
        __syncthreads();
        for (int i = tid; i < 16 * 256; i += 1024) (&whist[0][0])[i] = 0;
        __syncthreads();
        // pass 2: low-byte histogram inside the selected high-byte bin
        for (int i = tid; i < V8; i += 1024) {
            const u32x4 w = lv[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t k0 = order_key16(w[j] & 0xFFFFu, BF16), k1 = order_key16(w[j] >> 16, BF16);
                if ((k0 >> 8) == hb) atomicAdd(&whist[wave][k0 & 0xFFu], 1u);
                if ((k1 >> 8) == hb) atomicAdd(&whist[wave][k1 & 0xFFu], 1u);
            }
        }
        for (int i = (V8 << 3) + tid; i < V; i += 1024) {
            const uint32_t k = order_key16(logits[i], BF16);
            if ((k >> 8) == hb) atomicAdd(&whist[wave][k & 0xFFu], 1u);
        }
        __syncthreads();
        if (tid < 256) {
            unsigned int a = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) a += whist[w][tid];
            hist[tid] = a;
        }
        __syncthreads();
        select_bin(hist, suf, sel, need2, tid);
        if (tid == 0) sel[0] = (hb << 8) | sel[0];
        __syncthreads();
        pivot_key = sel[0];
    }
    // exponential race: argmax_i exp((x_i - max)/T) / q_i  over the kept set (the softmax
    // normaliser is common to all i and cannot change the argmax)
    const uint32_t seed = (uint32_t)rng_state[0], ctr = (uint32_t)rng_state[1];
    float best = -1.0f;
    int besti = 0x7FFFFFFF;
    auto consider = [&](const uint32_t b, const int i) {
        if (filter && order_key16(b, BF16) < pivot_key) return;
        const float pnum = expf((bits_to_float(b, BF16) - mx) * inv_temp);
        const float u = ((float)(hash3(seed, ctr, (uint32_t)i) >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float scv = pnum / (-logf(u));
        if (scv > best || (scv == best && i < besti)) { best = scv; besti = i; }
    };
    for (int i = tid; i < V8; i += 1024) {
        const u32x4 w = lv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            consider(w[j] & 0xFFFFu, i * 8 + 2 * j);
            consider(w[j] >> 16, i * 8 + 2 * j + 1);
        }
    }
    for (int i = (V8 << 3) + tid; i < V; i += 1024) consider(logits[i], i);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(best, d);
        const int oi = __shfl_xor(besti, d);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __syncthreads();
    if (lane == 0) { fred[wave] = best; ired[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (fred[w] > best || (fred[w] == best && ired[w] < besti)) { best = fred[w]; besti = ired[w]; }
        token_out[0] = besti;
        const unsigned long long c = rng_state[1];
        if (history && (long long)c < (long long)history_len) history[c] = besti;
        rng_state[1] = c + 1ull;
        if (pos_inout) pos_inout[0] = pos_inout[0] + 1;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct Config {
    int lpr, waves, split, unroll;
};

int g_num_cu = 0;
Config g_override = {0, 0, 0, 0};
unsigned long long* g_phase = nullptr;
int g_swizzle = 0;
int g_wave_local = 1;

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

size_t lds_bytes(int Z, int cap, int waves, int lpr, bool pair = false) {
    const int nch = (Z + 63) >> 6;
    return (size_t)nch * 8 + 128 + (size_t)cap * 4 + (size_t)waves * lpr * 8 * 4 * (pair ? 2 : 1) + 128;  // + int8 bias sums
}

int count_tiles(const Params& p, int bn) {
    int t = 0;
    for (int i = 0; i < p.nseg; ++i) t += (p.seg[i].ncols + bn - 1) / bn;
    return t;
}

// Launch geometry from (Z, columns, CU count).  Deterministic: no autotune at first call.
// Measured on MI355X (profiles/, scripts/tune_gemv.py): the kernel wants ONE 16-wave workgroup per
// CU (every workgroup repeats the compaction prologue, so more workgroups only add latency), and
// a single launch (split == 1) whenever the column tiles alone can occupy >= ~60 % of the CUs.
Config pick_config(int Z, int ncols_total, int nseg_tiles_hint) {
    (void)nseg_tiles_hint;
    const int ncu = g_num_cu > 0 ? g_num_cu : 256;
    Config c;
    c.waves = 16;
    c.unroll = 4;
    c.lpr = 0;
    c.split = 1;
    // widest tile whose tile count still covers most CUs -> no split-K, no second launch
    for (int lpr = 64; lpr >= 8; lpr >>= 1) {
        const int tiles = (ncols_total + lpr * 8 - 1) / (lpr * 8);
        if (tiles <= ncu + ncu / 8 && tiles * 5 >= ncu * 3) { c.lpr = lpr; break; }
    }
    if (c.lpr == 0) {
        const int t8 = (ncols_total + 63) / 64;
        if (t8 > ncu) {  // more 64-column tiles than CUs: widest tile that keeps >= ncu workgroups
            c.lpr = 8;
            for (int lpr = 64; lpr > 8; lpr >>= 1)
                if ((ncols_total + lpr * 8 - 1) / (lpr * 8) >= ncu) { c.lpr = lpr; break; }
        } else {  // few columns: 64-column tiles, split the kept rows to reach ~1 workgroup per CU.  With the
                  // wave-local compaction a slice is a set of 16-chunk rounds, so split <= rounds (and <= 8
                  // keeps the slab count small for consumers that re-read them); without it, 512-byte row
                  // segments and a deeper split measured best.
            const int rounds = (((Z + 63) >> 6) + 15) / 16;
            if (g_wave_local && rounds >= 2) {
                c.lpr = 8;
                const int tiles = (ncols_total + 63) / 64;
                int split = ncu / tiles;
                if (split > rounds) split = rounds;
                if (split > 8) split = 8;
                if (split < 1) split = 1;
                c.split = split;
            } else {
                c.lpr = ncols_total >= 2048 ? 32 : 8;
                const int tiles = (ncols_total + c.lpr * 8 - 1) / (c.lpr * 8);
                int split = ncu / tiles;
                const int max_by_rows = Z / (4 * c.waves * (64 / c.lpr));  // >= ~2 steps per workgroup at 50 %
                if (split > max_by_rows) split = max_by_rows;
                if (split < 1) split = 1;
                if (split > kMaxSplit) split = kMaxSplit;
                c.split = split;
            }
        }
    }
    if (g_override.lpr) c.lpr = g_override.lpr;
    if (g_override.waves) c.waves = g_override.waves;
    if (g_override.split) c.split = g_override.split;
    if (g_override.unroll) c.unroll = g_override.unroll;
    // LDS list capacity: stay inside the 64 KB a workgroup gets without opting in to more
    while ((size_t)((Z + c.split - 1) / c.split) * 4 > 40 * 1024 && c.split < kMaxSplit) ++c.split;
    return c;
}

template <int LPR, int WAVES, int U, int MODE, int KRT, bool PAIR>
hipError_t launch_gemv_k(const Params& p, int dtype, size_t lds, hipStream_t st) {
    const dim3 grid(p.ntiles * p.split), block(WAVES * 64);
    if (p.w8) {  // int8 weights: production geometry only (16 waves, unroll 4, tiles up to 256 columns)
        if constexpr (WAVES == 16 && U == 4 && LPR <= 32) {
            if (dtype == TEAL_BF16)
                hipLaunchKernelGGL((sparse_gemv_kernel<LPR, WAVES, U, true, MODE, KRT, PAIR, true>), grid, block, lds, st, p);
            else
                hipLaunchKernelGGL((sparse_gemv_kernel<LPR, WAVES, U, false, MODE, KRT, PAIR, true>), grid, block, lds, st, p);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if constexpr (U == 4 || (MODE == 0 && !PAIR)) {  // 16-bit fused variants are built for unroll 4 only
        if (dtype == TEAL_BF16)
            hipLaunchKernelGGL((sparse_gemv_kernel<LPR, WAVES, U, true, MODE, KRT, PAIR>), grid, block, lds, st, p);
        else
            hipLaunchKernelGGL((sparse_gemv_kernel<LPR, WAVES, U, false, MODE, KRT, PAIR>), grid, block, lds, st, p);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

// register-cache depth: smallest KRT with KRT * WAVES * 64 >= Z (16-wave production geometry);
// longer vectors use KRT = 16 plus the reload path (plain / silu-mul / masked producers only)
template <int LPR, int WAVES, int U, int MODE, bool PAIR>
hipError_t launch_gemv_m(const Params& p, int dtype, size_t lds, hipStream_t st) {
    if constexpr (WAVES == 16) {
        const int owned = p.krt ? p.krt : (((p.Z + 63) >> 6) + WAVES - 1) / WAVES;
        if (owned <= 4) return launch_gemv_k<LPR, WAVES, U, MODE, 4, PAIR>(p, dtype, lds, st);
        if (owned <= 8) return launch_gemv_k<LPR, WAVES, U, MODE, 8, PAIR>(p, dtype, lds, st);
        return launch_gemv_k<LPR, WAVES, U, MODE, 16, PAIR>(p, dtype, lds, st);
    } else {
        return launch_gemv_k<LPR, WAVES, U, MODE, 16, PAIR>(p, dtype, lds, st);
    }
}

template <int LPR, int WAVES, int U>
hipError_t launch_gemv_t(const Params& p, int dtype, size_t lds, hipStream_t st) {
    if (p.in.mode == 0 && !p.pair) return launch_gemv_m<LPR, WAVES, U, 0, false>(p, dtype, lds, st);
    if constexpr (WAVES == 16 && U == 4) {  // fused variants are built for the production geometry only
        if (p.pair) return p.in.mode == 1 ? launch_gemv_m<LPR, WAVES, U, 1, true>(p, dtype, lds, st) : hipErrorInvalidValue;
        if (p.in.mode == 1) return launch_gemv_m<LPR, WAVES, U, 1, false>(p, dtype, lds, st);
        if (p.in.mode == 2) return launch_gemv_m<LPR, WAVES, U, 2, false>(p, dtype, lds, st);
        if (p.in.mode == 3) return launch_gemv_m<LPR, WAVES, U, 3, false>(p, dtype, lds, st);
        if (p.in.mode == 4) return launch_gemv_m<LPR, WAVES, U, 4, false>(p, dtype, lds, st);
    }
    return hipErrorInvalidValue;
}

template <int LPR, int WAVES>
hipError_t launch_gemv_u(const Params& p, int dtype, size_t lds, int unroll, hipStream_t st) {
    switch (unroll) {
        case 4: return launch_gemv_t<LPR, WAVES, 4>(p, dtype, lds, st);
        case 8: return launch_gemv_t<LPR, WAVES, 8>(p, dtype, lds, st);
        default: return hipErrorInvalidValue;
    }
}

template <int LPR>
hipError_t launch_gemv_w(const Params& p, int dtype, size_t lds, const Config& c, hipStream_t st) {
    switch (c.waves) {
        case 8: return launch_gemv_u<LPR, 8>(p, dtype, lds, c.unroll, st);
        case 16: return launch_gemv_u<LPR, 16>(p, dtype, lds, c.unroll, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gemv(const Params& p, int dtype, size_t lds, const Config& c, hipStream_t st) {
    switch (c.lpr) {
        case 8: return launch_gemv_w<8>(p, dtype, lds, c, st);
        case 16: return launch_gemv_w<16>(p, dtype, lds, c, st);
        case 32: return launch_gemv_w<32>(p, dtype, lds, c, st);
        case 64: return launch_gemv_w<64>(p, dtype, lds, c, st);
        default: return hipErrorInvalidValue;
    }
}

// Common driver: fills geometry fields of `p` (segments' w/y/tau/ld/col0/ncols are set by the
// caller), launches the GEMV and, if needed, the ordered slab reduce.
int run_gemv(Params& p, int dtype, void* ws, size_t ws_bytes, bool to_ws, hipStream_t st,
             Config* used, bool few_slabs = false, bool interleave = false) {
    int total_cols = 0;
    for (int i = 0; i < p.nseg; ++i) total_cols += p.seg[i].ncols;
    Config c = pick_config(p.Z, total_cols, p.nseg);
    if (few_slabs && c.split > 1 && !g_override.split && !g_override.lpr) {
        // the consumer re-reads every slab in each of its workgroups: prefer narrow tiles and a
        // shallow split (64-column tiles, <= 8 slabs) over 512-byte row segments
        const int ncu = g_num_cu > 0 ? g_num_cu : 256;
        c.lpr = 8;
        const int tiles = (total_cols + 63) / 64;
        int split = ncu / tiles;
        if (split > 8) split = 8;
        if (split < 1) split = 1;
        c.split = split;
        while ((size_t)((p.Z + c.split - 1) / c.split) * 4 > 40 * 1024 && c.split < kMaxSplit) ++c.split;
        // a long vector can force more slices than CUs / tiles (LDS list capacity): then fill whole rounds of
        // workgroups (Llama-2-70B down, Z = 28672: 128 tiles x 3 slices = 1.5 rounds -> x 4 = 2 full rounds)
        while (tiles * c.split > ncu && (tiles * c.split) % ncu != 0 && c.split < 8) ++c.split;
    }
    if (p.w8) {
        // int8: 8 bytes per lane.  The stream is bound by cache-line REQUESTS per CU (measured: a 64-byte and a
        // 128-byte row segment cost the same, ~3.2 ns per row and CU), so a row segment should be a whole 128-byte
        // line = 16 lanes = 128 columns, as long as tiles x slices still cover most of the CUs
        if (c.lpr > 32) c.lpr = 32;
        if (!g_override.lpr && !g_override.split && !p.pair && c.lpr < 16) {
            const int ncu = g_num_cu > 0 ? g_num_cu : 256;
            const int tiles = (total_cols + 127) / 128;
            const int rounds = (((p.Z + 63) >> 6) + 15) / 16;
            int split = 1;
            if (to_ws) {  // slab output: the kept rows may be sliced (wave-local: <= rounds, interleaved slabs: <= 8)
                split = ncu / tiles;
                if (split > rounds) split = rounds;
                if (split > 8) split = 8;
                if (split < 1) split = 1;
            }
            if (tiles * split * 5 >= ncu * 3) {
                c.lpr = 16;
                c.split = split;
            }
        }
        for (int i = 0; i < (p.pair ? 2 : p.nseg); ++i)
            if (!p.seg[i].scale || (p.seg[i].ld & 7) || (p.seg[i].col0 & 7)) return TEAL_ERR_ARG;
    }
    if (p.in.mode != 0 || p.pair || p.w8) {  // fused / int8 variants exist for 16-wave workgroups
        c.waves = 16;
        c.unroll = 4;
        if ((p.in.mode == 1 || p.in.mode == 4) && p.Z > 16 * 64 * 16) return TEAL_ERR_SHAPE;  // register-resident producer
    }
    if (p.pair) {  // both matrices in one workgroup; the activation needs complete sums: no split-K
        c.split = 1;
        if ((size_t)(p.Z + 1) * 4 > 44 * 1024) return TEAL_ERR_SHAPE;
        if (!g_override.lpr) {
            const int ncu = g_num_cu > 0 ? g_num_cu : 256;
            const int n1 = p.seg[0].ncols;
            c.lpr = 8;
            for (int lpr = 64; lpr > 8; lpr >>= 1)  // widest tile that still gives >= 2/3 of the CUs a tile
                if (((n1 + lpr * 8 - 1) / (lpr * 8)) * 3 >= ncu * 2) { c.lpr = lpr; break; }
        }
    }
    const int bn = c.lpr * 8;
    int t = 0, off = 0;
    for (int i = 0; i < (p.pair ? 1 : p.nseg); ++i) {
        p.seg[i].tile0 = t;
        p.seg[i].ws_off = off;
        t += (p.seg[i].ncols + bn - 1) / bn;
        off += p.seg[i].ncols;
    }
    if (p.pair) p.nseg = 1;  // tile space = the gate columns; seg[1] rides along
    p.ntiles = t;
    p.split = c.split;
    p.ws_ld = off;
    p.cap = (p.Z + c.split - 1) / c.split + 1;
    p.wl = 0;
    p.sl = 0;
    p.krt = 0;
    if (g_wave_local && c.waves == 16) {
        const int nch = (p.Z + 63) >> 6;
        const int owned = (nch + 15) / 16;  // rounds of 16 chunks
        // element-wise producers (everything but the RMSNorm, which needs the whole vector in every workgroup)
        // cache only the rounds of the workgroup's slice
        const bool slice_local = p.in.mode != 1 && c.split > 1;
        const int need = slice_local ? (owned + c.split - 1) / c.split : owned;
        const int krt = need <= 4 ? 4 : (need <= 8 ? 8 : 16);
        const int capw = (slice_local ? need : (krt + c.split - 1) / c.split) * 64;  // entries one wave can own
        if (need <= krt && c.split <= owned && (size_t)16 * capw * 4 <= 40 * 1024) {
            p.wl = 1;
            p.sl = slice_local ? 1 : 0;
            p.krt = krt;
            p.cap = capw;
        }
    }
    p.to_ws = to_ws ? 1 : 0;
    p.ws = reinterpret_cast<float*>(ws);
    p.phase = g_phase;
    p.swizzle = g_swizzle;
    const size_t lds = lds_bytes(p.Z, p.wl ? p.cap * c.waves : p.cap, c.waves, c.lpr, p.pair != 0);
    if (lds > 64 * 1024) return TEAL_ERR_SHAPE;
    p.ws_il = (interleave && to_ws && c.split <= 8) ? 1 : 0;
    if (c.split > 1 || to_ws) {
        const size_t slabs = p.ws_il ? (size_t)((c.split + 3) & ~3) : (size_t)c.split;
        if (!ws || ws_bytes < slabs * off * sizeof(float)) return TEAL_ERR_WORKSPACE;
        if (!aligned16(ws)) return TEAL_ERR_ALIGN;
    }
    if (used) *used = c;
    if (launch_gemv(p, dtype, lds, c, st) != hipSuccess) return TEAL_ERR_LAUNCH;
    if (c.split > 1 && !to_ws) {
        const dim3 grid((off + 255) / 256), block(256);
        if (dtype == TEAL_BF16)
            hipLaunchKernelGGL((splitk_reduce_kernel<true>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<false>), grid, block, 0, st, p);
        if (hipGetLastError() != hipSuccess) return TEAL_ERR_LAUNCH;
    }
    return TEAL_OK;
}

int check_common(const void* x, const void* w, const void* y, int Z, int N, int dtype) {
    if (!x || !w || !y || Z <= 0 || N <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if ((N & 7) != 0 || Z > 65536) return TEAL_ERR_SHAPE;
    if (!aligned16(w) || (reinterpret_cast<uintptr_t>(x) & 1u) || (reinterpret_cast<uintptr_t>(y) & 1u))
        return TEAL_ERR_ALIGN;
    if (g_num_cu <= 0 && teal_init() <= 0) return TEAL_ERR_NO_DEVICE;
    return TEAL_OK;
}

}  // namespace

extern "C" {

int teal_version(void) { return 100; }

const char* teal_strerror(int code) {
    switch (code) {
        case TEAL_OK: return "ok";
        case TEAL_ERR_ARG: return "bad argument (null pointer or non-positive size)";
        case TEAL_ERR_DTYPE: return "unsupported dtype (0 = fp16, 1 = bf16)";
        case TEAL_ERR_SHAPE: return "unsupported shape (need N % 8 == 0, Z <= 65536, segment sizes % 8 == 0)";
        case TEAL_ERR_ALIGN: return "pointer not sufficiently aligned (weights/workspace 16 B)";
        case TEAL_ERR_WORKSPACE: return "split-K workspace missing or too small (teal_workspace_bytes)";
        case TEAL_ERR_LAUNCH: return "HIP kernel launch failed";
        case TEAL_ERR_NO_DEVICE: return "no HIP device available";
        case TEAL_ERR_CONFIG: return "invalid tuning override";
        default: return "unknown error";
    }
}

int teal_init(void) {
    if (g_num_cu > 0) return g_num_cu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return TEAL_ERR_NO_DEVICE;
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0)
        return TEAL_ERR_NO_DEVICE;
    g_num_cu = cu;
    return cu;
}

size_t teal_workspace_bytes(int Z, int N) {
    (void)Z;
    if (N <= 0) return 0;
    // kMaxSplit slabs of N columns; the fused gate|up GEMV uses two segments of N columns
    return (size_t)kMaxSplit * (size_t)N * 2 * sizeof(float);
}

int teal_set_tuning(int lanes_per_row, int waves, int split, int unroll) {
    auto in = [](int v, std::initializer_list<int> ok) {
        for (int o : ok) if (v == o) return true;
        return false;
    };
    if (!in(lanes_per_row, {0, 8, 16, 32, 64}) || !in(waves, {0, 8, 16}) ||
        !in(unroll, {0, 4, 8}) || split < 0 || split > kMaxSplit)
        return TEAL_ERR_CONFIG;
    g_override = {lanes_per_row, waves, split, unroll};
    return TEAL_OK;
}

int teal_set_wave_local(int on) {
    g_wave_local = on ? 1 : 0;
    return TEAL_OK;
}

int teal_set_swizzle(int on) {
    g_swizzle = on;
    return TEAL_OK;
}

int teal_set_phase_buffer(void* dev_u64) {
    g_phase = reinterpret_cast<unsigned long long*>(dev_u64);
    return TEAL_OK;
}

int teal_get_config(int Z, int N, int nseg, int* out) {
    if (!out || Z <= 0 || N <= 0) return TEAL_ERR_ARG;
    const Config c = pick_config(Z, N, nseg);
    out[0] = c.lpr;
    out[1] = c.waves;
    out[2] = c.split;
    out[3] = c.unroll;
    out[4] = ((N + c.lpr * 8 - 1) / (c.lpr * 8)) * c.split;
    return TEAL_OK;
}

int teal_compact(const void* x, float tau, int Z, int dtype, int32_t* idx_out, int32_t* count_out,
                 void* stream) {
    if (!x || !idx_out || !count_out || Z <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (Z > 65536) return TEAL_ERR_SHAPE;
    const int nch = (Z + 63) >> 6;
    const size_t lds = (size_t)nch * 8 + (size_t)(nch + 2) * 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const uint16_t* xp = reinterpret_cast<const uint16_t*>(x);
    if (dtype == TEAL_BF16)
        hipLaunchKernelGGL((compact_kernel<true>), dim3(1), dim3(1024), lds, st, xp, Z, tau, idx_out, count_out);
    else
        hipLaunchKernelGGL((compact_kernel<false>), dim3(1), dim3(1024), lds, st, xp, Z, tau, idx_out, count_out);
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_sparse_qkv_gemv(const void* x, const void* wT, void* y, float tau_q, float tau_k,
                         float tau_v, int Z, int N, int N_q, int N_kv, int dtype, void* ws,
                         size_t ws_bytes, void* stream) {
    return teal_sparse_qkv_gemv_ld(x, wT, N, y, tau_q, tau_k, tau_v, Z, N, N_q, N_kv, dtype, ws, ws_bytes, stream);
}

int teal_sparse_qkv_gemv_ld(const void* x, const void* wT, int ld, void* y, float tau_q, float tau_k,
                            float tau_v, int Z, int N, int N_q, int N_kv, int dtype, void* ws,
                            size_t ws_bytes, void* stream) {
    int rc = check_common(x, wT, y, Z, N, dtype);
    if (rc != TEAL_OK) return rc;
    if (ld < N || (ld & 7)) return TEAL_ERR_SHAPE;
    if (N_q < 0 || N_kv < 0 || N_q + N_kv > N || (N_q & 7) || (N_kv & 7)) return TEAL_ERR_SHAPE;
    Params p = {};
    p.x = x;
    p.Z = Z;
    const int widths[3] = {N_q, N_kv, N - N_q - N_kv};
    const float taus[3] = {tau_q, tau_k, tau_v};
    int col = 0, ns = 0;
    for (int i = 0; i < 3; ++i) {
        if (widths[i] > 0) {
            Seg& sgm = p.seg[ns++];
            sgm.w = wT;
            sgm.y = reinterpret_cast<uint16_t*>(y) + col;
            sgm.tau = taus[i];
            sgm.ld = ld;
            sgm.col0 = col;
            sgm.ncols = widths[i];
        }
        col += widths[i];
    }
    p.nseg = ns;
    return run_gemv(p, dtype, ws, ws_bytes, false, reinterpret_cast<hipStream_t>(stream), nullptr);
}

int teal_sparse_gemv(const void* x, const void* wT, void* y, float tau, int Z, int N, int dtype,
                     void* ws, size_t ws_bytes, void* stream) {
    return teal_sparse_qkv_gemv(x, wT, y, tau, tau, tau, Z, N, N, 0, dtype, ws, ws_bytes, stream);
}

int teal_sparse_qkv_gemv_i8(const void* x, const void* wqT, const void* scale, void* y, float tau_q, float tau_k,
                            float tau_v, int Z, int N, int N_q, int N_kv, int ld, int dtype, void* ws, size_t ws_bytes,
                            void* stream) {
    if (!scale || N_q <= 0 || N_kv < 0 || N_q + 2 * N_kv != N || ld < N) return TEAL_ERR_ARG;
    if ((N_q & 7) || (N_kv & 7) || (ld & 7)) return TEAL_ERR_SHAPE;
    int rc = check_common(x, wqT, y, Z, N, dtype);
    if (rc != TEAL_OK) return rc;
    teal_gemv_in_t in = {};
    in.mode = TEAL_IN_PLAIN;
    in.x = x;
    teal_gemv_out_t out = {};
    out.mode = TEAL_OUT_ROUNDED;
    out.weight_bits = 8;
    const float taus[3] = {tau_q, tau_k, tau_v};
    const int col0[3] = {0, N_q, N_q + N_kv}, ncols[3] = {N_q, N_kv, N_kv};
    out.nseg = N_kv > 0 ? 3 : 1;
    for (int i = 0; i < out.nseg; ++i) {
        out.w[i] = wqT;
        out.ld[i] = ld;
        out.col0[i] = col0[i];
        out.ncols[i] = ncols[i];
        out.tau[i] = taus[i];
        out.y[i] = reinterpret_cast<uint16_t*>(y) + col0[i];
        out.scale[i] = reinterpret_cast<const uint16_t*>(scale) + col0[i];
    }
    return teal_fused_gemv(&in, &out, Z, dtype, ws, ws_bytes, nullptr, stream);
}

int teal_dense_gemv(const void* x, const void* wT, void* y, int Z, int N, int dtype, void* ws,
                    size_t ws_bytes, void* stream) {
    // |x| > -inf keeps every finite and infinite activation; NaN propagates via nan_keeps.
    return teal_sparse_gemv(x, wT, y, -INFINITY, Z, N, dtype, ws, ws_bytes, stream);
}

int teal_sparse_gateup_silu(const void* x, const void* w1T, const void* w3T, void* h, float tau_gate,
                            float tau_up, int Z, int N, int dtype, void* ws, size_t ws_bytes,
                            void* stream) {
    int rc = check_common(x, w1T, h, Z, N, dtype);
    if (rc != TEAL_OK) return rc;
    if (!w3T) return TEAL_ERR_ARG;
    if (!aligned16(w3T)) return TEAL_ERR_ALIGN;
    Params p = {};
    p.x = x;
    p.Z = Z;
    p.nseg = 2;
    p.seg[0].w = w1T; p.seg[0].y = nullptr; p.seg[0].tau = tau_gate; p.seg[0].ld = N; p.seg[0].col0 = 0; p.seg[0].ncols = N;
    p.seg[1].w = w3T; p.seg[1].y = nullptr; p.seg[1].tau = tau_up;   p.seg[1].ld = N; p.seg[1].col0 = 0; p.seg[1].ncols = N;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    rc = run_gemv(p, dtype, ws, ws_bytes, true, st, nullptr);
    if (rc != TEAL_OK) return rc;
    const dim3 grid((N + 255) / 256), block(256);
    if (dtype == TEAL_BF16)
        hipLaunchKernelGGL((gateup_silu_epilogue_kernel<true>), grid, block, 0, st, p, h);
    else
        hipLaunchKernelGGL((gateup_silu_epilogue_kernel<false>), grid, block, 0, st, p, h);
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_fused_gemv(const teal_gemv_in_t* in, const teal_gemv_out_t* out, int Z, int dtype, void* ws,
                    size_t ws_bytes, int* nslabs_out, void* stream) {
    if (!in || !out || Z <= 0 || out->nseg < 1 || out->nseg > kMaxSeg) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (Z > 65536) return TEAL_ERR_SHAPE;
    if (g_num_cu <= 0 && teal_init() <= 0) return TEAL_ERR_NO_DEVICE;
    Params p = {};
    p.Z = Z;
    p.in.mode = in->mode;
    switch (in->mode) {
        case TEAL_IN_PLAIN:
        case TEAL_IN_SILU_MUL:
            if (!in->x) return TEAL_ERR_ARG;
            p.x = in->x;
            break;
        case TEAL_IN_ATTN_MERGE:
            if (!in->x || (in->att_head_dim != 64 && in->att_head_dim != 128) || Z % in->att_head_dim ||
                (in->att_nsplit != 0 && in->att_nsplit != 4 && in->att_nsplit != 8))
                return TEAL_ERR_ARG;
            p.in.att_ns = in->att_nsplit ? in->att_nsplit : 4;
            if (Z > (64 / p.in.att_ns) * 1024) return TEAL_ERR_SHAPE;  // one lane per (chunk, split)
            p.x = in->x;
            p.in.att = reinterpret_cast<const float*>(in->x);
            p.in.att_hd = in->att_head_dim;
            break;
        case TEAL_IN_MASKED:
            if (!in->x || !in->masks) return TEAL_ERR_ARG;
            p.x = in->x;
            p.in.masks = reinterpret_cast<const unsigned long long*>(in->masks);
            break;
        case TEAL_IN_RESID_NORM:
            if (!in->resid_in || !in->norm_weight || in->nslabs < 0 || (in->nslabs > 0 && !in->slabs)) return TEAL_ERR_ARG;
            if (in->resid_out == in->resid_in && !in->row_index) return TEAL_ERR_ARG;  // must ping-pong
            p.x = in->resid_in;
            p.in.resid_in = in->resid_in;
            p.in.row_index = in->row_index;
            p.in.slabs = in->slabs;
            p.in.nslabs = in->nslabs;
            p.in.slabs_il = in->slabs_interleaved ? 1 : 0;
            if (p.in.slabs_il && in->nslabs > 8) return TEAL_ERR_ARG;
            p.in.norm_w = in->norm_weight;
            p.in.resid_out = in->resid_out;
            p.in.eps = in->eps;
            break;
        default: return TEAL_ERR_ARG;
    }
    p.nseg = out->nseg;
    for (int i = 0; i < out->nseg; ++i) {
        if (!out->w[i] || out->ncols[i] <= 0 || (out->ncols[i] & 7) || (out->col0[i] & 7) || (out->ld[i] & 7)) return TEAL_ERR_SHAPE;
        if (!aligned16(out->w[i])) return TEAL_ERR_ALIGN;
        if (out->mode == TEAL_OUT_ROUNDED && !out->y[i]) return TEAL_ERR_ARG;
        p.seg[i].w = out->w[i];
        p.seg[i].y = out->y[i];
        p.seg[i].tau = out->tau[i];
        p.seg[i].ld = out->ld[i];
        p.seg[i].col0 = out->col0[i];
        p.seg[i].ncols = out->ncols[i];
        p.seg[i].scale = out->scale[i];
    }
    if (out->weight_bits != 0 && out->weight_bits != 16 && out->weight_bits != 8) return TEAL_ERR_ARG;
    p.w8 = out->weight_bits == 8 ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Config used = {};
    int rc;
    if (out->mode == TEAL_OUT_PAIR_SILU) {
        // seg 0 = gate (w1), seg 1 = up (w3), same shape; y[0] receives h = silu(gate) * up
        if (out->nseg != 2 || in->mode != TEAL_IN_RESID_NORM || out->ncols[0] != out->ncols[1] || !out->y[0]) return TEAL_ERR_ARG;
        p.pair = 1;
        p.mask_out = reinterpret_cast<unsigned long long*>(out->mask_out);
        p.mask_tau = out->mask_tau;
        rc = run_gemv(p, dtype, ws, ws_bytes, false, st, &used, false);
    } else if (out->mode == TEAL_OUT_SLABS) {
        if (!out->slabs) return TEAL_ERR_ARG;
        rc = run_gemv(p, dtype, out->slabs, out->slabs_bytes, true, st, &used, true, out->slabs_interleaved != 0);
        if (rc == TEAL_OK && out->slabs_interleaved && !p.ws_il) return TEAL_ERR_CONFIG;  // > 8 slices cannot interleave
    } else if (out->mode == TEAL_OUT_ROUNDED) {
        rc = run_gemv(p, dtype, ws, ws_bytes, false, st, &used, false);
    } else {
        return TEAL_ERR_ARG;
    }
    if (rc == TEAL_OK && nslabs_out) *nslabs_out = used.split;
    return rc;
}

int teal_decode_attention_masked(const void* qkv, const void* rope, const int32_t* pos, void* k_cache, void* v_cache,
                                 void* y, void* mask_out, float mask_tau, int n_head, int n_kv_head, int head_dim,
                                 int max_seq, int dtype, void* stream) {
    if (!qkv || !rope || !pos || !k_cache || !v_cache || !y) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if ((head_dim != 64 && head_dim != 128) || n_head <= 0 || n_kv_head <= 0 || n_head % n_kv_head || max_seq <= 0)
        return TEAL_ERR_SHAPE;
    // 16 waves per head: a wave load covers whole cache rows, one pass covers 1024 positions
    const int nt = 1024;
    const size_t lds = (size_t)(3 * head_dim + 2 * (nt / 64) + (nt / 64) * head_dim + max_seq) * sizeof(float);
    if (lds > 64 * 1024) return TEAL_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
    const dim3 grid(n_head), block(nt);
    auto* q = reinterpret_cast<const uint16_t*>(qkv);
    auto* r = reinterpret_cast<const uint16_t*>(rope);
    auto* kc = reinterpret_cast<uint16_t*>(k_cache);
    auto* vc = reinterpret_cast<uint16_t*>(v_cache);
    auto* yo = reinterpret_cast<uint16_t*>(y);
    auto* mo = reinterpret_cast<unsigned long long*>(mask_out);
#define TEAL_ATT(BF, NTV, HDV) hipLaunchKernelGGL((decode_attention_kernel<BF, NTV, HDV>), grid, block, lds, st, q, r, pos, kc, vc, yo, mo, mask_tau, n_head, n_kv_head, max_seq, scale, g_phase)
#define TEAL_ATT_HD(BF, NTV) do { if (head_dim == 128) TEAL_ATT(BF, NTV, 128); else TEAL_ATT(BF, NTV, 64); } while (0)
    if (dtype == TEAL_BF16) TEAL_ATT_HD(true, 1024);
    else TEAL_ATT_HD(false, 1024);
#undef TEAL_ATT_HD
#undef TEAL_ATT
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

static int attention_split_impl(const void* qkv, const float* qkv_slabs, int qkv_nslabs, const void* rope, const int32_t* pos,
                                void* k_cache, void* v_cache, void* y, void* mask_out, float mask_tau, int n_head,
                                int n_kv_head, int head_dim, int max_seq, int nsplit, void* partials, size_t partials_bytes,
                                int dtype, void* stream) {
    if ((!qkv && !qkv_slabs) || !rope || !pos || !k_cache || !v_cache || !partials) return TEAL_ERR_ARG;
    if (qkv_slabs && (qkv_nslabs < 1 || qkv_nslabs > 8 || !aligned16(qkv_slabs))) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if ((head_dim != 64 && head_dim != 128) || n_head <= 0 || n_kv_head <= 0 || n_head % n_kv_head || max_seq <= 0 ||
        nsplit < 1 || nsplit > 64)
        return TEAL_ERR_SHAPE;
    if (partials_bytes < (size_t)n_head * nsplit * (head_dim + 2) * sizeof(float)) return TEAL_ERR_WORKSPACE;
    const int chunk_max = (max_seq + nsplit - 1) / nsplit;
    // bandwidth of one workgroup = bytes in flight / latency: long shares get 16 waves (the whole K and V
    // share of up to 256 rows is requested up front), short ones 4 waves (cheaper barriers)
    const int nt = chunk_max > 128 ? 1024 : 256;
    const size_t lds = (size_t)(3 * head_dim + 2 * (nt / 64) + (nt / 64) * head_dim + chunk_max) * sizeof(float);
    if (lds > 64 * 1024) return TEAL_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
    auto* q = reinterpret_cast<const uint16_t*>(qkv);
    auto* r = reinterpret_cast<const uint16_t*>(rope);
    auto* kc = reinterpret_cast<uint16_t*>(k_cache);
    auto* vc = reinterpret_cast<uint16_t*>(v_cache);
    auto* pw = reinterpret_cast<float*>(partials);
    const dim3 grid(n_head * nsplit), block(nt);
#define TEAL_ATTS(BF, HDV, NTV) hipLaunchKernelGGL((decode_attention_split_kernel<BF, HDV, NTV>), grid, block, lds, st, q, r, pos, kc, vc, pw, n_head, n_kv_head, max_seq, nsplit, chunk_max, scale, qkv_slabs, qkv_nslabs)
#define TEAL_ATTS_NT(BF, HDV) do { if (nt == 1024) TEAL_ATTS(BF, HDV, 1024); else TEAL_ATTS(BF, HDV, 256); } while (0)
    if (dtype == TEAL_BF16) { if (head_dim == 128) TEAL_ATTS_NT(true, 128); else TEAL_ATTS_NT(true, 64); }
    else { if (head_dim == 128) TEAL_ATTS_NT(false, 128); else TEAL_ATTS_NT(false, 64); }
#undef TEAL_ATTS_NT
#undef TEAL_ATTS
    if (hipGetLastError() != hipSuccess) return TEAL_ERR_LAUNCH;
    if (!y) return TEAL_OK;  // partials only: the consumer merges (TEAL_IN_ATTN_MERGE)
    auto* yo = reinterpret_cast<uint16_t*>(y);
    auto* mo = reinterpret_cast<unsigned long long*>(mask_out);
    if (dtype == TEAL_BF16)
        hipLaunchKernelGGL((decode_attention_merge_kernel<true>), dim3(n_head), dim3(128), 0, st, pw, yo, mo, mask_tau, head_dim, nsplit);
    else
        hipLaunchKernelGGL((decode_attention_merge_kernel<false>), dim3(n_head), dim3(128), 0, st, pw, yo, mo, mask_tau, head_dim, nsplit);
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_decode_attention_split(const void* qkv, const void* rope, const int32_t* pos, void* k_cache, void* v_cache,
                                void* y, void* mask_out, float mask_tau, int n_head, int n_kv_head, int head_dim,
                                int max_seq, int nsplit, void* partials, size_t partials_bytes, int dtype, void* stream) {
    if (!qkv) return TEAL_ERR_ARG;
    return attention_split_impl(qkv, nullptr, 0, rope, pos, k_cache, v_cache, y, mask_out, mask_tau, n_head, n_kv_head,
                                head_dim, max_seq, nsplit, partials, partials_bytes, dtype, stream);
}

int teal_decode_attention_split_slabs(const float* qkv_slabs, int qkv_nslabs, const void* rope, const int32_t* pos,
                                      void* k_cache, void* v_cache, void* y, void* mask_out, float mask_tau, int n_head,
                                      int n_kv_head, int head_dim, int max_seq, int nsplit, void* partials,
                                      size_t partials_bytes, int dtype, void* stream) {
    if (!qkv_slabs) return TEAL_ERR_ARG;
    return attention_split_impl(nullptr, qkv_slabs, qkv_nslabs, rope, pos, k_cache, v_cache, y, mask_out, mask_tau, n_head,
                                n_kv_head, head_dim, max_seq, nsplit, partials, partials_bytes, dtype, stream);
}

int teal_decode_attention(const void* qkv, const void* rope, const int32_t* pos, void* k_cache, void* v_cache,
                          void* y, int n_head, int n_kv_head, int head_dim, int max_seq, int dtype, void* stream) {
    return teal_decode_attention_masked(qkv, rope, pos, k_cache, v_cache, y, nullptr, 0.0f, n_head, n_kv_head, head_dim,
                                        max_seq, dtype, stream);
}

int teal_sample_topk(const void* logits, int vocab, int dtype, int top_k, float temperature, void* rng_state,
                     int32_t* token_out, int32_t* pos_inout, int32_t* history, int history_len, void* stream) {
    if (!logits || !rng_state || !token_out || vocab <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (!aligned16(logits)) return TEAL_ERR_ALIGN;
    const float inv_temp = 1.0f / fmaxf(temperature, 1e-5f);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto* lg = reinterpret_cast<const uint16_t*>(logits);
    auto* rs = reinterpret_cast<unsigned long long*>(rng_state);
    if (dtype == TEAL_BF16)
        hipLaunchKernelGGL((sample_topk_kernel<true>), dim3(1), dim3(1024), 0, st, lg, vocab, top_k, inv_temp, rs, token_out, pos_inout, history, history_len);
    else
        hipLaunchKernelGGL((sample_topk_kernel<false>), dim3(1), dim3(1024), 0, st, lg, vocab, top_k, inv_temp, rs, token_out, pos_inout, history, history_len);
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

}  // extern "C"
