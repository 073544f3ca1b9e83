// teal_gemv_int4.hip — sparse GEMV over int4 group-quantised weights (second half).
//
// The reference ships int4-g32/64/128 weight-only linears for its DENSE gpt-fast path only
// (gpt-fast/quantize.py:58-162 group q-params / quantise / dequantise, :359-443 handler, :483-526 WeightOnlyInt4Linear,
// whose forward is a CUDA-only tinygemm op) and lists quantised TEAL as missing (README.md:110).  Semantics restated:
//     w[n][m] = (q[n][m] - 8) * scale[m / G][n] + zero[m / G][n]        q in 0..15, scale / zero bf16, G = group size
//     y[n]    = sum over kept m of x[m] * w[n][m]                        kept: float32(|x[m]|) > float32(tau), strict
// scales_and_zeros [Z / G][N][2] bf16 exactly as the reference stores them (quantize.py:79-93).
//
// Weight layout (ours to choose — the reference's packed layout is tinygemm's): the image of W^T by ROW PAIRS,
// wq[Z / 2][ldb] bytes; dword g of pair-row p holds columns 4g .. 4g+3 of row 2p in its low half (nibble j = column 4g+j)
// and of row 2p+1 in its high half.  One mask-and-or under an fp16 exponent then yields the half2 (1024 + q[2p][c],
// 1024 + q[2p+1][c]) and one v_dot2_f32_f16 against (x[2p], x[2p+1]) does two multiply-adds in fp32: ~1.2 VALU instructions
// per multiply-add where a row-per-dword layout needs ~2.6 (unpack, convert, fma) — and this kernel is bound by the VALU,
// not by HBM: 4 waves per SIMD share it, and at 4 bits a byte of weights carries 4x the arithmetic of fp16.  A pair is
// fetched when EITHER row is kept (the other row's x enters as 0, which is exact): at 50 % sparsity 75 % of the pairs,
// i.e. 1.5x the bytes of a row-granular gather — bytes this kernel has to spare.
//
// Kernel: one 16-wave workgroup = one 128-column tile x one slice of the 32-row units.  A wave owns units; per unit it
// ballots the keep mask, compacts the kept pairs into its LDS lists, deals them to its four 16-lane groups (a lane = 8
// columns = 8 bytes of a 128-byte pair-row segment), accumulates A = sum x * (bias + q) and X = sum x in fp32 and applies
// scale / zero ONCE per (unit, column):  y += scale * (A - (bias + 8) X) + zero * X  — the group parameters cost 32 bytes
// per lane and unit instead of per row.  Split-K over units is folded into the one launch by arrival tickets (the last
// slice of a tile sums the partials in slice order: deterministic, no atomics on the data).  No MFMA on purpose.
//
// The kernel takes the producers of the fused decode step like the 16-bit one (teal_gemv_fast.h): MODE 1 residual + fp32
// slabs -> RMSNorm (every workgroup recomputes the vector into LDS: the norm needs all of it), MODE 2 silu(gate) * up and
// MODE 4 split-KV attention merge (element-wise: each wave builds the 32 activations of its own units in registers) — and
// leaves either rounded outputs or its fp32 split-K partials as slabs for the next launch's producer, so an int4 layer is
// 5 launches too.
#include "teal_common.h"

#include <limits.h>

// Same rule as the 16-bit GEMV units (teal_gemv_kernel.h): no implicit contraction — every fused multiply-add below is an
// explicit fmaf(), so the slab path and the rounded path, and this unit against the merge producers elsewhere, cannot drift
// by an ulp when the compiler changes its mind.
#pragma clang fp contract(off)
#include <stdio.h>

namespace teal {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
constexpr int kPhaseRow = 32;  // uint64 stamps per workgroup (as in teal_gemv_kernel.h)

struct I4Seg {
    const unsigned char* wq;  // first byte of the segment's columns in row 0
    const uint16_t* sz;       // (scale, zero) of the segment's first column in group 0
    uint16_t* y;              // rounded output of the segment (null when the launch leaves slabs)
    int ldb, szld;            // pair-row stride of wq in bytes; columns per row of sz
    int tile0;                // first 128-column tile of the segment
    float tau;
};

struct I4Args {
    I4Seg seg[3];
    int nseg, Z, gshift;      // gshift: log2 of the group size
    // producer
    const uint16_t* x;        // MODE 0: x[Z]; MODE 2: gate[Z] | up[Z]; MODE 1: residual (table with row_index)
    const int* row_index;
    const float* slabs;       // MODE 1: interleaved fp32 partials [Z][(nslabs + 3) & ~3] folded into the residual
    int nslabs;
    const uint16_t* norm_w;
    float eps;
    uint16_t* resid_out;
    const float* att;         // MODE 4: partials [Z / hd][ns][hd + 2]
    int att_hd, att_ns;
    // split-K partials / slabs: element (column c, slice s) at ws[c * ws_es + s * ws_ss]
    float* ws;
    int ws_es, ws_ss;
    unsigned* ticket;         // null: split == 1, or the partials stay as slabs for the consumer
    unsigned long long* phase;  // PHASE instantiations: kPhaseRow stamps per workgroup (teal_set_phase_buffer)
};

// A wave handles its 32-row units in PASSES of four: the activations of the pass first; the ballots compact the kept row
// PAIRS of each unit into the wave's LDS lists (pair index, and the two activations as a half2 with a dropped row's at 0)
// with mbcnt ranks; then each of the wave's four 16-lane groups takes ONE unit of the pass — its own group parameters, its
// own accumulators, no cross-lane traffic until the end of the kernel — and EVERY pair-row load of the pass is issued before
// the first is consumed.  (Four groups sharing the pairs of one unit would each fetch that unit's 32 bytes of parameters
// per lane: the launch was then bound by the CU's 64 bytes / clock of vector-memory issue, half of it those parameters.)
// scale / zero are applied per UNIT with the parameters of the unit's group — y += scale * (A_u - c X_u) + zero * X_u is
// linear in the units of a group — so units are independent whatever the group size.  fp16 activations: nibbles 1 and 3 of
// a half are used where they lie, as 1024 + 16 q (no shift); their columns are rescaled at the flush.  bf16 activations:
// bias 128 (7 mantissa bits), every nibble shifted down.
// PHASE (measurement builds only): thread 0 stamps [0] entry, [1] arguments in registers, [2] producer done (MODE 1), and for
// its wave's first pass [3] activations ready, [4] list written, [5] every load issued, [6] first unit consumed, [7] pass
// done; [8] all passes done, [9] past the reduce barrier, [10] outputs stored.  100 MHz wall clock (scripts/int4_phase.py).
// KIND, by the units a launch leaves a wave (host: fused_gemv_i4):
//   1 SHARE  one or two (the 7B wo projection over 256 workgroups: one) — the four lane groups SHARE each unit, pairs dealt
//            round-robin, instead of three of them idling while one walks 12 pairs on its own;
//   0 GROUP  more: passes of four units, one unit per lane group, loads issued only for the pairs that exist.
// (A software-pipelined form of GROUP — pass p + 1's loads in flight while pass p is consumed, every lane issuing exactly 16
// loads and running 16 steps per pass so that the loop body is straight-line and the compiler's vmcnt waits exact — measured
// SLOWER: 494 against 537 tok/s on Llama-2-7B, 66.4 against 89.1 on Llama-2-70B, int4-g32 @ 50 %; 128 VGPRs with spills,
// 17 % more loads and steps than the pairs that exist.  Starting the odd waves half a pass late, so that one half of the
// chip's waves loads while the other computes: 88.1 / 86.2 against 89.4 tok/s (70B), 533 / 523 against 539 (7B).  A ring of 8
// loads per lane that never drains between passes, the next pass's lists built while the current one streams (ISA checked:
// exact progressive vmcnt waits): 561 against 578, 97.4 against 102.9.  Passes of eight units, two per lane group (32 loads in
// flight per lane, the 7B gate|up launch in one pass): 567 against 578, 99.4 against 102.8.  None is kept: overlap inside a wave is not what
// bounds the launch — padding every unit to 16 steps costs more than the overlap returns.)
constexpr int kI4Group = 0, kI4Share = 1;
template <bool BF16, int MODE, int KIND, bool PHASE = false>
__global__ __launch_bounds__(1024) void sparse_gemv_int4_kernel(const I4Args a) {
    constexpr bool SHARE = KIND == kI4Share;
    constexpr int WAVES = 16, BN = 128, UP = SHARE ? 2 : 4;  // UP: units per pass
    unsigned long long t_entry = 0;
    if constexpr (PHASE) t_entry = wall_clock64();
    extern __shared__ __align__(16) uint16_t xs[];  // MODE 1: the normalised activation vector, Z entries
    __shared__ float red[WAVES * BN];
    // 16 slots per unit: ascending; SHARE: rank k at slot (k & 3) * 4 + (k >> 2), a lane group's four entries contiguous
    __shared__ __align__(16) uint8_t list_i[WAVES][UP * 16];   // pair index inside the unit (one byte: a lane's 16 in one read)
    __shared__ __align__(16) uint32_t list_x[WAVES][UP * 16];  // (x[2p], x[2p + 1]) as 16-bit halves, a dropped row's = 0
    __shared__ float wsum[WAVES];
    __shared__ float tflag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, slice = blockIdx.y, split = gridDim.y;
    const int rs = lane >> 4, cl = lane & 15;  // pair slot of the wave, 8-byte column slot of the tile
    const int Z = a.Z;
    bool first_pass = true;
    auto stamp = [&](const int i) {
        if constexpr (PHASE) {
            if (a.phase && tid == 0 && first_pass) a.phase[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kPhaseRow + i] = wall_clock64();
        }
    };
    // ONE batch of scalar loads for every kernel argument, up front (an "s" input forces the value into an SGPR here):
    // left to the compiler they are fetched where first used — a chain of dependent scalar-cache misses through the
    // prologue and again in the epilogue
#define TEAL_I4_SEG_ARGS(g) "s"((g).wq), "s"((g).sz), "s"((g).y), "s"((g).ldb), "s"((g).szld), "s"((g).tile0), "s"((g).tau)
    asm volatile("" ::TEAL_I4_SEG_ARGS(a.seg[0]), TEAL_I4_SEG_ARGS(a.seg[1]), TEAL_I4_SEG_ARGS(a.seg[2]));
    asm volatile("" ::"s"(a.nseg), "s"(a.Z), "s"(a.gshift), "s"(a.x), "s"(a.row_index), "s"(a.slabs), "s"(a.nslabs), "s"(a.norm_w),
                 "s"(a.eps), "s"(a.resid_out), "s"(a.att), "s"(a.att_hd), "s"(a.att_ns), "s"(a.ws), "s"(a.ws_es), "s"(a.ws_ss),
                 "s"(a.ticket));
#undef TEAL_I4_SEG_ARGS
    if constexpr (PHASE) {
        if (a.phase && tid == 0) a.phase[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kPhaseRow] = t_entry;
    }
    stamp(1);

    // ---- producer -------------------------------------------------------------------------------------------------------
    if constexpr (MODE == 1) {
        // h = resid + round(sum slabs);  x = round(round(h * rstd) * w)          gpt-fast/model.py:158-161, 289-291
        // KM elements per thread (4: Z <= 4096, 8: Z <= 8192, else 16); every load is unconditional (index clamped) and
        // leaves in one batch (KM = 16: per block of 4 elements)
        const uint16_t* resid = a.x;
        if (a.row_index) resid += (size_t)a.row_index[0] * (size_t)Z;  // embedding row of the current token
        const uint32_t stride = (uint32_t)(a.nslabs + 3) & ~3u;
        const int nslabs = a.nslabs;
        const bool writer = a.resid_out && tile == 0 && slice == 0;
        auto produce = [&](auto km_tag) {
            constexpr int KM = decltype(km_tag)::value, KB = KM <= 8 ? KM : 4;
            uint32_t wb[KM];
#pragma unroll
            for (int k = 0; k < KM; ++k) wb[k] = a.norm_w[min(tid + (k << 10), Z - 1)];
            float ss = 0.0f;
#pragma unroll
            for (int k0 = 0; k0 < KM; k0 += KB) {
                uint32_t rb[KB];
                f32x4 v0[KB], v1[KB];
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const uint32_t m = (uint32_t)min(tid + ((k0 + j) << 10), Z - 1);
                    rb[j] = resid[m];
                    if (nslabs > 0) v0[j] = *reinterpret_cast<const f32x4*>(a.slabs + m * stride);
                    if (nslabs > 4) v1[j] = *reinterpret_cast<const f32x4*>(a.slabs + m * stride + 4);
                }
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int m = tid + ((k0 + j) << 10);
                    float r = bits_to_float(rb[j], BF16);
                    if (nslabs > 0) {  // slab order 0, 1, 2, ... (the order of the ordered reduce); an absent slab adds 0.0f
                        float sacc = 0.0f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) sacc += (q < nslabs) ? v0[j][q] : 0.0f;
                        if (nslabs > 4) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) sacc += (4 + q < nslabs) ? v1[j][q] : 0.0f;
                        }
                        const float yv = bits_to_float(float_to_bits<BF16>(sacc), BF16);
                        r = bits_to_float(float_to_bits<BF16>(r + yv), BF16);
                    }
                    if (m < Z) {
                        xs[m] = float_to_bits<BF16>(r);  // exact: r is a 16-bit value
                        ss = fmaf(r, r, ss);
                    }
                }
            }
            ss = wave_sum_f(ss);
            if (lane == 0) wsum[wave] = ss;
            __syncthreads();
            float tot = lane < WAVES ? wsum[lane] : 0.0f;
            tot = wave_sum_f(tot);
            const float rstd = rsqrtf(tot / (float)Z + a.eps);
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const int m = tid + (k << 10);
                if (m < Z) {
                    const uint32_t hb = xs[m];  // this thread's own element
                    const float xn = bits_to_float(float_to_bits<BF16>(bits_to_float(hb, BF16) * rstd), BF16);
                    xs[m] = float_to_bits<BF16>(xn * bits_to_float(wb[k], BF16));
                    if (writer) a.resid_out[m] = (uint16_t)hb;
                }
            }
        };
        if (Z <= 4096) produce(std::integral_constant<int, 4>{});
        else if (Z <= 8192) produce(std::integral_constant<int, 8>{});
        else produce(std::integral_constant<int, 16>{});
        __syncthreads();
    }

    stamp(2);
    // ---- this workgroup's column tile: segment, threshold, weight image ---------------------------------------------------
    // (selected field by field from kernel arguments already in SGPRs: indexing a.seg[] by a run-time s would be a
    // dependent scalar load)
    int s = 0;
    if (a.nseg > 1 && tile >= a.seg[1].tile0) s = 1;
    if (a.nseg > 2 && tile >= a.seg[2].tile0) s = 2;
#define TEAL_I4_SEG(f) (s == 0 ? a.seg[0].f : (s == 1 ? a.seg[1].f : a.seg[2].f))
    const unsigned char* seg_wq = TEAL_I4_SEG(wq);
    const uint16_t* seg_sz = TEAL_I4_SEG(sz);
    uint16_t* seg_y = TEAL_I4_SEG(y);
    const int ldb = TEAL_I4_SEG(ldb), szld = TEAL_I4_SEG(szld), seg_tile0 = TEAL_I4_SEG(tile0);
    const float tau = TEAL_I4_SEG(tau);
#undef TEAL_I4_SEG
    const uint32_t scol = (uint32_t)(tile - seg_tile0) * BN + cl * 8;  // the lane's first column inside the segment
    // a pair-row holds one byte per column.  Addresses = workgroup-uniform base (scalar registers) + a 32-bit lane offset:
    // one VGPR and one multiply-add per load instead of a 64-bit pointer each (the host bounds the image below 4 GB)
    const unsigned char* wtile = seg_wq + (uint32_t)(tile - seg_tile0) * BN;
    const uint32_t lane_off = (uint32_t)cl * 8u;
    const uint32_t uldb = (uint32_t)ldb;
    const uint16_t* szb = seg_sz + (size_t)scol * 2;
    const int nunits = Z >> 5;
    uint8_t* li = list_i[wave];
    uint32_t* lx = list_x[wave];
    constexpr uint32_t kBias = BF16 ? 0x43004300u : 0x64006400u;  // half2 (128, 128) bf16 / (1024, 1024) fp16
    constexpr uint32_t kOnes = BF16 ? 0x3F803F80u : 0x3C003C00u;  // half2 (1, 1)
    auto dot2 = [](const uint32_t w2, const uint32_t x2, const float acc) {
        if constexpr (BF16) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w2), __builtin_bit_cast(bf16x2_t, x2), acc, false);
        else return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, w2), __builtin_bit_cast(f16x2, x2), acc, false);
    };
    float total[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) total[k] = 0.0f;
    // 16 multiply-adds of a lane: 8 columns x the two rows of a pair
    auto mac16 = [&](const u32x2 dd, const uint32_t x2, float (&A)[8], float& X) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t w = dd[h];
            if constexpr (BF16) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) A[4 * h + jj] = dot2(((w >> (4 * jj)) & 0x000F000Fu) | kBias, x2, A[4 * h + jj]);
            } else {
                const uint32_t w8 = w >> 8;
                A[4 * h + 0] = dot2((w & 0x000F000Fu) | kBias, x2, A[4 * h + 0]);
                A[4 * h + 1] = dot2((w & 0x00F000F0u) | kBias, x2, A[4 * h + 1]);  // 1024 + 16 q
                A[4 * h + 2] = dot2((w8 & 0x000F000Fu) | kBias, x2, A[4 * h + 2]);
                A[4 * h + 3] = dot2((w8 & 0x00F000F0u) | kBias, x2, A[4 * h + 3]);  // 1024 + 16 q
            }
        }
        X = dot2(kOnes, x2, X);
    };
    // scale / zero of one unit: sum x (q - 8) from the biased accumulators; a dead unit has X = A = 0
    auto flush = [&](const float (&A)[8], const float X, const u32x4 s0, const u32x4 s1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t pr = k < 4 ? s0[k] : s1[k - 4];  // bf16 pair: scale (low half), zero (high half)
            const float sc = __uint_as_float(pr << 16), zr = __uint_as_float(pr & 0xFFFF0000u);
            float t;
            if constexpr (BF16) t = A[k] - 136.0f * X;                         // (128 + q) - 136
            else if (k & 1) t = fmaf(A[k], 0.0625f, -72.0f * X);               // ((1024 + 16 q) - 1024) / 16 - 8
            else t = A[k] - 1032.0f * X;                                       // (1024 + q) - 1032
            total[k] += fmaf(sc, t, zr * X);
        }
    };
    // unit u belongs to slice u % split, and inside the slice to wave (u / split) % 16
    const int ustride = split * WAVES;
    // phases 1 and 2 of a pass whose first unit is u0: activations, ballots, the wave's lists (li, lx); cnt[i] = kept pairs
    auto front = [&](const int u0, uint8_t* li, uint32_t* lx, int (&cnt)[UP], bool (&live)[UP]) {
        // ---- 1. activations of the pass ----------------------------------------------------------------------------------------
        // The element-wise producers (MODE 2, MODE 4) run HERE, per unit, in the registers of the wave that owns the unit:
        // a workgroup touches only its own slice of gate | up (of the attention partials), not the whole vector.
        uint32_t xb[UP];
        uint32_t m_el[UP];  // the lane's element of unit i (lanes 32..63 mirror 0..31)
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            const int u = u0 + i * ustride;
            live[i] = u < nunits;  // wave-uniform
            const int uu = live[i] ? u : u0;
            m_el[i] = (uint32_t)(uu << 5) + (lane & 31);
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < UP; ++i) xb[i] = a.x[m_el[i]];
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < UP; ++i) xb[i] = xs[m_el[i]];
        } else if constexpr (MODE == 2) {
            // x = round(round(silu(gate)) * up)                                      gpt-fast/model.py:258-259
            uint32_t gb[UP], ub[UP];
#pragma unroll
            for (int i = 0; i < UP; ++i) {  // all gate / up loads of the pass first
                gb[i] = a.x[m_el[i]];
                ub[i] = a.x[(uint32_t)Z + m_el[i]];
            }
#pragma unroll
            for (int i = 0; i < UP; ++i) {
                const float gt = bits_to_float(gb[i], BF16);
                const float sl = bits_to_float(float_to_bits<BF16>(gt / (1.0f + expf(-gt))), BF16);
                xb[i] = float_to_bits<BF16>(sl * bits_to_float(ub[i], BF16));
            }
        } else {
            // attention output merged from the split-KV partials {max, sum, o[hd]} per (head, split); the arithmetic of the
            // 16-bit kernel's merge producer: pairwise-tree sum of the rescaled denominators, fmaf chain over the splits
            const int hd = a.att_hd, hs = hd + 2;
            auto merge = [&](auto ns_tag, auto i0_tag, auto n_tag) {
                constexpr int NS = decltype(ns_tag)::value, I0 = decltype(i0_tag)::value, NU = decltype(n_tag)::value;
                float2 st[NU][NS];
                float ov[NU][NS];
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    const int m = (int)m_el[I0 + j];
                    const int h = m / hd, d = m - h * hd;
                    const float* b = a.att + (size_t)h * NS * hs;
#pragma unroll
                    for (int q = 0; q < NS; ++q) {
                        st[j][q] = *reinterpret_cast<const float2*>(b + q * hs);
                        ov[j][q] = b[q * hs + 2 + d];
                    }
                }
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    float M = st[j][0].x;
#pragma unroll
                    for (int q = 1; q < NS; ++q) M = fmaxf(M, st[j][q].x);
                    float f[NS], t[NS];
#pragma unroll
                    for (int q = 0; q < NS; ++q) {
                        f[q] = st[j][q].y > 0.0f ? expf(st[j][q].x - M) : 0.0f;
                        t[q] = st[j][q].y * f[q];
                    }
                    float Ls = (t[0] + t[1]) + (t[2] + t[3]);
                    if constexpr (NS == 8) Ls = Ls + ((t[7] + t[6]) + (t[5] + t[4]));
                    float Os = 0.0f;
#pragma unroll
                    for (int q = 0; q < NS; ++q) Os = fmaf(ov[j][q], f[q] / Ls, Os);
                    xb[I0 + j] = float_to_bits<BF16>(Os);
                }
            };
            using I = std::integral_constant<int, 0>;
            if (a.att_ns == 8) {  // 24 registers of partials per unit: two units at a time
                merge(std::integral_constant<int, 8>{}, I{}, std::integral_constant<int, 2>{});
                if constexpr (UP == 4) merge(std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
            } else {
                merge(std::integral_constant<int, 4>{}, I{}, std::integral_constant<int, UP>{});
            }
        }
        stamp(3);
        // ---- 2. ballots -> kept row pairs of every unit in the wave's lists (16 slots per unit, ascending) -----------------
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            const float v = bits_to_float(xb[i], BF16);
            const bool keep = keep_rule(v, tau) || (v != v);
            const uint32_t mask = live[i] ? (uint32_t)__ballot(keep) : 0u;  // lanes 32..63 mirror 0..31
            const uint32_t pm = (mask | (mask >> 1)) & 0x55555555u;          // bit 2p: pair p has a kept row
            const uint32_t xm = keep ? xb[i] : 0u;
            const uint32_t xn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xm, 0xB1, 0xf, 0xf, false);  // lane ^ 1
            if (lane < 32 && ((pm >> lane) & 1u)) {  // even lanes of kept pairs
                const uint32_t rank = __builtin_amdgcn_mbcnt_lo(pm, 0u);
                const uint32_t slot = (uint32_t)i * 16u + (SHARE ? (rank & 3u) * 4u + (rank >> 2) : rank);
                li[slot] = (uint8_t)(lane >> 1);
                lx[slot] = xm | (xn << 16);
            }
            cnt[i] = __popc(pm);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int u0 = slice + split * wave; u0 < nunits; u0 += ustride * UP) {
        int cnt[UP];  // kept pairs of unit i (wave-uniform)
        bool live[UP];
        front(u0, li, lx, cnt, live);
        stamp(4);
        if constexpr (SHARE) {
            // ---- 3s. the four lane groups share each unit: group rs takes the pairs of rank rs, rs + 4, ... (<= 4 steps per unit)
            u32x4 sz0[UP], sz1[UP];
#pragma unroll
            for (int i = 0; i < UP; ++i) {
                const int uu = live[i] ? u0 + i * ustride : u0;  // wave-uniform
                const u32x4* szp = reinterpret_cast<const u32x4*>(szb + (size_t)((uu << 5) >> a.gshift) * szld * 2);
                sz0[i] = szp[0];
                sz1[i] = szp[1];
            }
            u32x2 d[UP][4];
#pragma unroll
            for (int i = 0; i < UP; ++i) {
                const uint32_t pidx = *reinterpret_cast<const uint32_t*>(li + i * 16 + rs * 4);  // the group's four pair indices
                const uint32_t row_off = (uint32_t)((live[i] ? u0 + i * ustride : u0) << 4) * uldb + lane_off;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d[i][r] = u32x2{0u, 0u};
                    if (4 * r < cnt[i]) {          // wave-uniform: this step has pairs
                        if (4 * r + rs < cnt[i])   // (a slot past the count holds a stale index: never dereferenced)
                            d[i][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wtile + (row_off + ((pidx >> (8 * r)) & 0xFFu) * uldb)));
                    }
                }
            }
            stamp(5);
            // ---- 4s. arithmetic, unit by unit; scale / zero once per (unit, column) — linear, so every group applies them to
            //          its own share and the shares meet in the reduce at the end of the kernel
#pragma unroll
            for (int i = 0; i < UP; ++i) {
                if (i == 1) stamp(6);
                float A[8], X = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k) A[k] = 0.0f;
                const u32x4 xx4 = *reinterpret_cast<const u32x4*>(lx + i * 16 + rs * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (4 * r < cnt[i]) {
                        const uint32_t x2 = (4 * r + rs < cnt[i]) ? xx4[r] : 0u;  // a group without a pair adds 0
                        mac16(d[i][r], x2, A, X);
                    }
                }
                flush(A, X, sz0[i], sz1[i]);
            }
        } else {
            // ---- 3. 16-lane group rs takes unit rs of the pass: its group parameters (32 bytes per lane, once) and every
            //         pair-row load of its unit, one pair per step, all issued before the first is consumed ----------------------
            const int cntv = rs == 0 ? cnt[0] : (rs == 1 ? cnt[1] : (rs == 2 ? cnt[2] : cnt[3]));
            const int maxcnt = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));  // wave-uniform step count
            const int uv = min(u0 + rs * ustride, nunits - 1);                 // (a dead unit has no pairs: any valid address)
            const u32x4* szp = reinterpret_cast<const u32x4*>(szb + (size_t)((uv << 5) >> a.gshift) * szld * 2);
            const u32x4 sz0 = szp[0], sz1 = szp[1];
            const uint32_t row_off = (uint32_t)(uv << 4) * uldb + lane_off;  // first pair-row of the group's unit
            u32x2 d[16];
            {
                const u32x4 pidx = *reinterpret_cast<const u32x4*>(li + rs * 16);  // the group's 16 pair indices
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    d[r] = u32x2{0u, 0u};
                    if (r < maxcnt) {     // wave-uniform
                        if (r < cntv)     // (a slot past the count holds a stale index: never dereferenced)
                            d[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wtile + (row_off + ((pidx[r >> 2] >> (8 * (r & 3))) & 0xFFu) * uldb)));
                    }
                }
            }
            stamp(5);
            // ---- 4. arithmetic; scale / zero once per (unit, column) ---------------------------------------------------------------
            {
                float A[8], X = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k) A[k] = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q == 1) stamp(6);
                    if (q * 4 < maxcnt) {
                        const u32x4 xx4 = *reinterpret_cast<const u32x4*>(lx + rs * 16 + q * 4);
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int r = q * 4 + rr;
                            if (r < maxcnt) {
                                const uint32_t x2 = (r < cntv) ? xx4[rr] : 0u;  // a group past its count adds 0
                                    mac16(d[r], x2, A, X);
                            }
                        }
                    }
                }
                    flush(A, X, sz0, sz1);
            }
        }
        stamp(7);
        first_pass = false;
        __builtin_amdgcn_wave_barrier();  // the lists are rewritten by the next pass
    }
    first_pass = true;
    stamp(8);
    // ---- reduce: the four row groups of the wave, then the waves in fixed order -------------------------------------------
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        total[k] = xor_add<16>(total[k]);
        total[k] = xor_add<32>(total[k]);
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) red[wave * BN + lane * 8 + k] = total[k];
    }
    __syncthreads();
    stamp(9);
    const uint32_t c = (uint32_t)tile * BN + tid;  // column in the concatenation of the segments (slab / partial index)
    uint16_t* yp = seg_y ? seg_y + (uint32_t)(tile - seg_tile0) * BN + tid : nullptr;
    float sum = 0.0f;
    if (tid < BN) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sum += red[w * BN + tid];
        if (!a.ws) *yp = float_to_bits<BF16>(sum);
        else if (a.ticket) __hip_atomic_store(&a.ws[c * (uint32_t)a.ws_es + (uint32_t)slice * a.ws_ss], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.ws[c * (uint32_t)a.ws_es + (uint32_t)slice * a.ws_ss] = sum;  // slabs for the next launch's producer
    }
    if (a.ticket) {  // arrival tickets: see gemv_fast_kernel
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(&a.ticket[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tflag = (t == (unsigned)split - 1u) ? 1.0f : 0.0f;
        }
        __syncthreads();
        if (tflag != 0.0f && tid < BN) {
            float acc = 0.0f;
            for (int sl = 0; sl < split; ++sl)
                acc += __hip_atomic_load(&a.ws[c * (uint32_t)a.ws_es + (uint32_t)sl * a.ws_ss], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *yp = float_to_bits<BF16>(acc);
            if (tid == 0) __hip_atomic_store(&a.ticket[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (PHASE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(10);
    }
}

template <bool BF16, int KIND, bool PHASE>
static hipError_t launch_i4(const I4Args& a, int mode, dim3 grid, size_t lds, hipStream_t st) {
    const dim3 block(1024);
    switch (mode) {
        case TEAL_IN_PLAIN: hipLaunchKernelGGL((sparse_gemv_int4_kernel<BF16, 0, KIND, PHASE>), grid, block, 0, st, a); break;
        case TEAL_IN_RESID_NORM: hipLaunchKernelGGL((sparse_gemv_int4_kernel<BF16, 1, KIND, PHASE>), grid, block, lds, st, a); break;
        case TEAL_IN_SILU_MUL: hipLaunchKernelGGL((sparse_gemv_int4_kernel<BF16, 2, KIND, PHASE>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((sparse_gemv_int4_kernel<BF16, 4, KIND, PHASE>), grid, block, 0, st, a); break;
    }
    return hipGetLastError();
}

// One launch over int4 weights: [producer] -> mask + compaction -> gathered GEMV over every segment (teal_fused_gemv with
// weight_bits = 4, and the plain entry point below).  split-K factor: enough slices to cover the CUs, at most 8, every
// wave of every slice owning at least one unit; without a destination for the partials (no prepared workspace and no
// slabs requested) the launch keeps every unit of a tile in one workgroup.
int fused_gemv_i4(const teal_gemv_in_t* in, const teal_gemv_out_t* out, int Z, int dtype, void* ws, size_t ws_bytes,
                  int* nslabs_out, hipStream_t st) {
    const int G = out->groupsize;
    if (G != 32 && G != 64 && G != 128 && G != 256) return TEAL_ERR_ARG;
    if (Z <= 0 || (Z % G) || Z > 65536) return TEAL_ERR_SHAPE;
    if (out->nseg < 1 || out->nseg > 3) return TEAL_ERR_ARG;
    if (out->mode != TEAL_OUT_ROUNDED && out->mode != TEAL_OUT_SLABS) return TEAL_ERR_ARG;  // no paired gate|up form
    DeviceCtx* dc = device_ctx();
    if (!dc) return TEAL_ERR_NO_DEVICE;
    I4Args a = {};
    a.nseg = out->nseg; a.Z = Z; a.gshift = G == 32 ? 5 : (G == 64 ? 6 : (G == 128 ? 7 : 8));
    int ntiles = 0, N = 0;
    for (int i = 0; i < out->nseg; ++i) {
        const int nc = out->ncols[i], c0 = out->col0[i], ldb = out->ld[i], szld = out->scale_ld[i];
        if (!out->w[i] || !out->scale[i] || nc <= 0 || (nc % 128) || c0 < 0 || (c0 & 7)) return TEAL_ERR_SHAPE;
        if (ldb < c0 + nc || (ldb & 7) || szld < c0 + nc) return TEAL_ERR_SHAPE;  // a pair-row: one byte per column
        if ((size_t)(Z / 2) * (size_t)ldb >= ((size_t)1 << 32)) return TEAL_ERR_SHAPE;  // 32-bit offsets inside an image
        if ((reinterpret_cast<uintptr_t>(out->w[i]) & 7u) || !aligned16(out->scale[i])) return TEAL_ERR_ALIGN;
        if (out->mode == TEAL_OUT_ROUNDED && !out->y[i]) return TEAL_ERR_ARG;
        I4Seg& sg = a.seg[i];
        sg.wq = reinterpret_cast<const unsigned char*>(out->w[i]) + c0;
        sg.sz = reinterpret_cast<const uint16_t*>(out->scale[i]) + (size_t)c0 * 2;
        sg.y = out->mode == TEAL_OUT_ROUNDED ? reinterpret_cast<uint16_t*>(out->y[i]) : nullptr;
        sg.ldb = ldb; sg.szld = szld; sg.tile0 = ntiles; sg.tau = out->tau[i];
        ntiles += nc / 128;
        N += nc;
    }
    size_t lds = 0;
    switch (in->mode) {
        case TEAL_IN_PLAIN:
            if (!in->x) return TEAL_ERR_ARG;
            a.x = reinterpret_cast<const uint16_t*>(in->x);
            break;
        case TEAL_IN_SILU_MUL:
            if (!in->x) return TEAL_ERR_ARG;
            a.x = reinterpret_cast<const uint16_t*>(in->x);
            break;
        case TEAL_IN_ATTN_MERGE:
            if (!in->x || (in->att_head_dim != 64 && in->att_head_dim != 128) || Z % in->att_head_dim ||
                (in->att_nsplit != 0 && in->att_nsplit != 4 && in->att_nsplit != 8))
                return TEAL_ERR_ARG;
            a.att = reinterpret_cast<const float*>(in->x);
            a.att_hd = in->att_head_dim;
            a.att_ns = in->att_nsplit ? in->att_nsplit : 4;
            break;
        case TEAL_IN_RESID_NORM:
            if (!in->resid_in || !in->norm_weight || in->nslabs < 0 || (in->nslabs > 0 && !in->slabs)) return TEAL_ERR_ARG;
            if (in->resid_out == in->resid_in && !in->row_index) return TEAL_ERR_ARG;  // must ping-pong
            if (in->nslabs > 0 && (!in->slabs_interleaved || in->nslabs > 8 || !aligned16(in->slabs))) return TEAL_ERR_ARG;
            if (Z > 16384) return TEAL_ERR_SHAPE;  // 16 elements per thread, 32 KB of LDS
            a.x = reinterpret_cast<const uint16_t*>(in->resid_in);
            a.row_index = in->row_index;
            a.slabs = in->slabs;
            a.nslabs = in->nslabs;
            a.norm_w = reinterpret_cast<const uint16_t*>(in->norm_weight);
            a.eps = in->eps;
            a.resid_out = reinterpret_cast<uint16_t*>(in->resid_out);
            lds = (size_t)Z * 2;
            break;
        default: return TEAL_ERR_ARG;  // TEAL_IN_MASKED: no int4 form
    }
    const int nunits = Z / 32;
    int split = dc->num_cu / ntiles;
    if (split > 8) split = 8;
    if (split * 16 > nunits) split = nunits / 16;  // every wave of every slice owns at least one 32-row unit
    if (split < 1) split = 1;
    if (out->mode == TEAL_OUT_SLABS) {
        if (!out->slabs || !aligned16(out->slabs) || ws_prepared(out->slabs, out->slabs_bytes)) return TEAL_ERR_ARG;
        a.ws = out->slabs;
        if (out->slabs_interleaved) { a.ws_es = (split + 3) & ~3; a.ws_ss = 1; }
        else { a.ws_es = 1; a.ws_ss = N; }
        if (out->slabs_bytes < (size_t)(out->slabs_interleaved ? a.ws_es : split) * N * sizeof(float)) return TEAL_ERR_WORKSPACE;
    } else if (split > 1) {
        if (!ws_prepared(ws, ws_bytes) || ntiles > kTicketTiles) split = 1;
        else {
            a.ws_es = (split + 3) & ~3; a.ws_ss = 1;
            if (ws_bytes - kWsHeaderBytes < (size_t)a.ws_es * N * sizeof(float)) return TEAL_ERR_WORKSPACE;
            a.ws = ws_slabs(ws);
            a.ticket = ws_tickets(ws);
        }
    }
    const dim3 grid(ntiles, split);
    // a wave's units: up to two -> its lane groups share each unit; more -> passes of four, a unit per lane group
    const int upw = (nunits + split * 16 - 1) / (split * 16);
    const int kind = upw <= 2 ? kI4Share : kI4Group;
    hipError_t e;
#define TEAL_I4_LAUNCH(BF, PH) \
    (kind == kI4Share ? launch_i4<BF, kI4Share, PH>(a, in->mode, grid, lds, st) : launch_i4<BF, kI4Group, PH>(a, in->mode, grid, lds, st))
#ifdef TEAL_DIAGNOSTICS
    if (g_phase && dtype == TEAL_F16) {  // measurement: the stamping instantiation (fp16 only; libteal_hip_diag.so)
        a.phase = phase_next_region();
        e = TEAL_I4_LAUNCH(false, true);
    } else
#endif
    if (dtype == TEAL_BF16) {
        e = TEAL_I4_LAUNCH(true, false);
    } else {
        e = TEAL_I4_LAUNCH(false, false);
    }
#undef TEAL_I4_LAUNCH
    if (e != hipSuccess) return TEAL_ERR_LAUNCH;
    if (out->desc || kDiagnostics) {
        char d[160];
        snprintf(d, sizeof d, "sparse_gemv_int4_kernel<%s,%d,%d,false> grid (%d,%d) x 1024", dtype == TEAL_BF16 ? "true" : "false",
                 in->mode, kind, ntiles, split);
        publish_desc(d, out->desc, out->desc_bytes);
    }
    if (nslabs_out) *nslabs_out = split;
    return TEAL_OK;
}

}  // namespace teal

using namespace teal;

extern "C" int teal_sparse_qkv_gemv_i4(const void* x, const void* wq, const void* scales_and_zeros, void* y, float tau_q,
                                       float tau_k, float tau_v, int Z, int N, int N_q, int N_kv, int ldb, int groupsize,
                                       int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !wq || !scales_and_zeros || !y || Z <= 0 || N <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (groupsize != 32 && groupsize != 64 && groupsize != 128 && groupsize != 256) return TEAL_ERR_ARG;
    if ((N % 128) || (Z % groupsize) || Z > 65536 || ldb < N || (ldb & 7)) return TEAL_ERR_SHAPE;
    if (N_q <= 0 || N_kv < 0 || N_q + 2 * N_kv != N || (N_q % 128) || (N_kv % 128)) return TEAL_ERR_SHAPE;
    teal_gemv_in_t in = {};
    in.mode = TEAL_IN_PLAIN;
    in.x = x;
    teal_gemv_out_t out = {};
    out.nseg = N_kv > 0 ? 3 : 1;
    const int c0[3] = {0, N_q, N_q + N_kv}, nc[3] = {N_q, N_kv, N_kv};
    const float tau[3] = {tau_q, tau_k, tau_v};
    for (int i = 0; i < out.nseg; ++i) {
        out.w[i] = wq; out.scale[i] = scales_and_zeros;
        out.ld[i] = ldb; out.scale_ld[i] = N;
        out.col0[i] = c0[i]; out.ncols[i] = nc[i]; out.tau[i] = tau[i];
        out.y[i] = reinterpret_cast<uint16_t*>(y) + c0[i];
    }
    out.mode = TEAL_OUT_ROUNDED;
    out.weight_bits = 4;
    out.groupsize = groupsize;
    return fused_gemv_i4(&in, &out, Z, dtype, ws, ws_bytes, nullptr, reinterpret_cast<hipStream_t>(stream));
}
