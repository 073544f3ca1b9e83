// teal_gemv_kernel.h — the sparse GEMV kernel template and its launch dispatch.  Included by the four
// teal_gemv_w*_*.hip translation units, each of which instantiates one (weight width, dtype) quadrant.
//
// Replaces (reference tree FasterDecoding/TEAL @ 2024-10-22):
//   kernels/sparse_gemv.py:50-83    splitk_sparse_gemv_kernel   -> sparse_gemv_kernel<>
//   kernels/sparse_gemv.py:152-194  qkv_kernel                  -> sparse_gemv_kernel<> (3 segments)
//   kernels/sparse_gemv.py:8-12     init_to_zero("Y") memset    -> gone (no accumulation into Y)
//   kernels/sparse_gemv.py:83       fp16 tl.atomic_add split-K  -> fp32 slabs + ordered reduce
//
// Design (DESIGN.md §3.1 has the long form):
//   * One 16-wave workgroup = one column tile (LPR lanes x 8 columns = BN) x one slice of the kept rows.
//   * Producer (MODE): the activation is loaded — or computed: residual + slabs -> RMSNorm, silu(gate)*up,
//     split-KV attention merge — into registers, all loads in flight before the first use.
//   * Mask + compaction, wave-local: a wave ballots its own 64-element chunks (fp32(|x|) > fp32(tau), strict;
//     kernels/sparse_gemv.py:75) and writes the surviving (row:16 | x:16) pairs into its private LDS list
//     (rank = mbcnt of the ballot).  No cross-wave scan, no barrier before the first weight load.  Fallback for
//     vectors beyond the register cache: masks -> LDS, DPP prefix scan, one workgroup-wide list, even shares.
//   * Stream: a wave walks its list 64/LPR rows at a time; a lane issues U independent non-temporal 16-byte
//     (int8: 8-byte) loads, two batches in flight, fp32 accumulators; no LDS staging (a GEMV has no reuse).
//   * Reduce: shuffle across the row groups of a wave, LDS across waves in fixed order; one rounding, or an
//     fp32 slab per slice summed in slice order by the consumer.  No atomics: bit-reproducible.
//   * HBM-bound skinny GEMV: no MFMA on purpose (north_star).
#pragma once
#include "teal_common.h"

// Floating-point contraction is OFF in the GEMV translation units: every fused multiply-add below is written as fmaf().
// hipcc's default (-ffp-contract=fast) lets the compiler fuse `a * b + c` wherever it sees fit, and it saw fit differently
// in this kernel and in the lean kernel (teal_gemv_fast.h) for the same source line of the attention-merge producer: the
// activation handed to the wo projection then differed in the last place for some inputs, i.e. the two kernels — which
// are specified to be bit-identical — were not (round 2's "wo slabs DIFF"; found by tests/test_soak.py).  With the
// contraction explicit, what is written is what runs, in both.
#pragma clang fp contract(off)

namespace teal {

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


// optional per-workgroup phase timestamps (constant 100 MHz clock, comparable across CUs).  Layout: 32 uint64 per
// workgroup: [0] kernel entry (taken before any kernel argument is used), [1] kernel arguments in registers,
// [2] activation ready, [3] list ready, [4] first weight batch consumed (wave 0), [5] wave 0 done streaming,
// [6] after the reduce barrier, [7] done, [12] hw id | xcc, [13] grid | waves << 32, [16 + w] end of stream of wave w
constexpr int kPhaseRow = 32;
__device__ __forceinline__ void stamp(const Params& p, int phase) {
    if (p.phase && threadIdx.x == 0) p.phase[(size_t)blockIdx.x * kPhaseRow + phase] = wall_clock64();
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

    const unsigned long long t_entry = wall_clock64();  // before the first use of a kernel argument
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
    const int tile = blockIdx.x % p.ntiles;
    const int slice = blockIdx.x / p.ntiles;
    // (Block b runs on XCD b % 8, so every XCD only ever touches one residue class mod 8 of column tiles; an XOR swizzle of the
    // low tile bits that spreads each XCD over all residues measured neutral once the row stride is padded — DESIGN.md 3.1 — and is
    // gone since round 5.)

    int s = 0;
    if (p.nseg > 1 && tile >= p.seg[1].tile0) s = 1;
    if (p.nseg > 2 && tile >= p.seg[2].tile0) s = 2;
    const Seg sg = p.seg[s];
    const int tcol0 = (tile - sg.tile0) * BN;  // first column of the tile inside the segment

    const uint16_t* __restrict__ x = reinterpret_cast<const uint16_t*>(p.x);
    if (p.phase && threadIdx.x == 0) p.phase[(size_t)blockIdx.x * kPhaseRow] = t_entry;
    stamp(p, 1);
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
    // with an element-wise producer): a workgroup only ever needs the chunks of ITS slice, so register k caches
    // chunk slice + split * (wave + WAVES k) instead — 1/split of the loads, and vectors up to split * 16 rounds fit the cache
    // (balanced over the slices to within one chunk, the rule of gemv_fast_kernel: slice-local chunk c belongs to slice c mod
    //  split and wave (c div split) mod WAVES; otherwise wave w holds chunks w + WAVES k and owns those with (k + w) mod split
    //  == slice)
    auto chunk_of = [&](const int k) { return p.sl ? slice + p.split * (wave + WAVES * k) : wave + WAVES * k; };
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
            if (p.in.gate_act) return float_to_bits<BF16>(gt * up);  // the producer applied silu (act_seg0)
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
            ss = fmaf(r, r, ss);  // explicit: see the contract(off) note at the top of this file
        }
        stamp(p, 8);   // loads back, row sums formed
        ss = wave_sum_f(ss);
        if (lane == 0) sumsq[wave] = ss;
        __syncthreads();
        stamp(p, 9);   // past the producer barrier
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
                for (int q = 0; q < NS; ++q) Os = fmaf(ov[k][q], __int_as_float(__builtin_amdgcn_readlane(cw, NS * k + q)), Os);
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
            const float sl = p.in.gate_act ? gt : bits_to_float(float_to_bits<BF16>(gt / (1.0f + expf(-gt))), BF16);
            xr[k] = (m < Z) ? (uint32_t)float_to_bits<BF16>(sl * bits_to_float(ub[k], BF16)) : 0u;
        }
    } else {
#pragma unroll
        for (int k = 0; k < KR; ++k) xr[k] = x[mcl[k]];
    }
    stamp(p, 2);
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
        int base = 0, kmod = (p.sl || p.split == 1) ? 0 : wave % p.split;
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
        stamp(p, 3);
    } else {
            if constexpr (MODE != 3) {
            int mycnt = 0;
        #pragma unroll
            for (int k = 0; k < GREG * PER; ++k) {  // whole groups of 64 chunks only (the reload loop takes the rest)
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
            __syncthreads();
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
        bool first_done = false;
        if (fa) issue(wa, w2a, xa, eb);
        while (fa) {
            int ebn = eb + STEP;
            const bool fb = full(ebn);
            if (fb) issue(wb, w2b, xb, ebn);
            consume(wa, w2a, xa);
            if (p.phase && !first_done) { first_done = true; stamp(p, 4); }
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

    stamp(p, 5);
    if (p.phase && lane == 0) p.phase[(size_t)blockIdx.x * kPhaseRow + 16 + (wave & 15)] = wall_clock64();  // per-wave end of stream
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
    stamp(p, 6);
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
                reinterpret_cast<uint16_t*>(sg.y)[c] = (p.act0 && s == 0) ? silu_bits<BF16>(sum) : float_to_bits<BF16>(sum);
            } else if (p.ws_il) {
                p.ws[(size_t)(sg.ws_off + c) * ((p.split + 3) & ~3) + slice] = sum;
            } else {
                p.ws[(size_t)slice * p.ws_ld + sg.ws_off + c] = sum;
            }
        }
    }
    stamp(p, 7);
    if (p.phase && threadIdx.x == 0) {
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        p.phase[(size_t)blockIdx.x * kPhaseRow + 12] = ((unsigned long long)hwid << 32) | xcc;
        p.phase[(size_t)blockIdx.x * kPhaseRow + 13] = ((unsigned long long)WAVES << 32) | gridDim.x;
    }
}


// ---- launch dispatch: runtime (lanes per row, waves, unroll, producer mode, register-cache depth) -> template
//      instantiation, for one (activation dtype, weight width) quadrant -------------------------------------

// which instantiations exist: 16-wave workgroups, unroll 4 (the production geometry), every tile width / producer /
// cache depth.  The 8-wave and unroll-8 variants of round 1 were sweep-only (measured no better; 4-wave workgroups
// worse: the launch is bound by HBM and by instruction issue in the prologue, not by dispatch) and are no longer built.
template <bool W8, int LPR, int WAVES, int U, int MODE, int KRT, bool PAIR>
constexpr bool variant_built() {
    if (WAVES != 16 || U != 4) return false;
    return W8 ? LPR <= 32 : true;
}

template <bool BF16, bool W8, int LPR, int WAVES, int U, int MODE, int KRT, bool PAIR>
hipError_t launch_gemv_k(const Params& p, size_t lds, hipStream_t st) {
    const dim3 grid(p.ntiles * p.split), block(WAVES * 64);
    if constexpr (variant_built<W8, LPR, WAVES, U, MODE, KRT, PAIR>()) {
        hipLaunchKernelGGL((sparse_gemv_kernel<LPR, WAVES, U, BF16, MODE, KRT, PAIR, W8>), grid, block, lds, st, p);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

// register-cache depth: smallest KRT with KRT * WAVES * 64 >= Z, or what the wave-local slice needs (p.krt);
// longer vectors use KRT = 16 plus the reload path
template <bool BF16, bool W8, int LPR, int WAVES, int U, int MODE, bool PAIR>
hipError_t launch_gemv_m(const Params& p, size_t lds, hipStream_t st) {
    if constexpr (WAVES == 16) {
        const int owned = p.krt ? p.krt : (((p.Z + 63) >> 6) + WAVES - 1) / WAVES;
        if (owned <= 4) return launch_gemv_k<BF16, W8, LPR, WAVES, U, MODE, 4, PAIR>(p, lds, st);
        if (owned <= 8) return launch_gemv_k<BF16, W8, LPR, WAVES, U, MODE, 8, PAIR>(p, lds, st);
    }
    return launch_gemv_k<BF16, W8, LPR, WAVES, U, MODE, 16, PAIR>(p, lds, st);
}

template <bool BF16, bool W8, int LPR, int WAVES, int U>
hipError_t launch_gemv_t(const Params& p, size_t lds, hipStream_t st) {
    if (p.in.mode == 0 && !p.pair) return launch_gemv_m<BF16, W8, LPR, WAVES, U, 0, false>(p, lds, st);
    if (p.pair) return p.in.mode == 1 ? launch_gemv_m<BF16, W8, LPR, WAVES, U, 1, true>(p, lds, st) : hipErrorInvalidValue;
    if (p.in.mode == 1) return launch_gemv_m<BF16, W8, LPR, WAVES, U, 1, false>(p, lds, st);
    if (p.in.mode == 2) return launch_gemv_m<BF16, W8, LPR, WAVES, U, 2, false>(p, lds, st);
    if (p.in.mode == 3) return launch_gemv_m<BF16, W8, LPR, WAVES, U, 3, false>(p, lds, st);
    if (p.in.mode == 4) return launch_gemv_m<BF16, W8, LPR, WAVES, U, 4, false>(p, lds, st);
    return hipErrorInvalidValue;
}

template <bool BF16, bool W8, int LPR, int WAVES>
hipError_t launch_gemv_u(const Params& p, size_t lds, int unroll, hipStream_t st) {
    switch (unroll) {
        case 4: return launch_gemv_t<BF16, W8, LPR, WAVES, 4>(p, lds, st);
        default: return hipErrorInvalidValue;
    }
}

template <bool BF16, bool W8, int LPR>
hipError_t launch_gemv_w(const Params& p, size_t lds, const Config& c, hipStream_t st) {
    switch (c.waves) {
        case 16: return launch_gemv_u<BF16, W8, LPR, 16>(p, lds, c.unroll, st);
        default: return hipErrorInvalidValue;
    }
}

template <bool BF16, bool W8>
hipError_t launch_gemv_q(const Params& p, size_t lds, const Config& c, hipStream_t st) {
    switch (c.lpr) {
        case 8: return launch_gemv_w<BF16, W8, 8>(p, lds, c, st);
        case 16: return launch_gemv_w<BF16, W8, 16>(p, lds, c, st);
        case 32: return launch_gemv_w<BF16, W8, 32>(p, lds, c, st);
        case 64: return launch_gemv_w<BF16, W8, 64>(p, lds, c, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace teal
