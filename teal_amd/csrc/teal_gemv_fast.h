// teal_gemv_fast.h — the lean sparse GEMV kernel of the fused decode step (16-bit weights, whole 64-element chunks,
// whole column tiles).  Same algorithm and the same arithmetic, in the same order, as sparse_gemv_kernel
// (teal_gemv_kernel.h, which stays the general kernel: ragged shapes, int8 weights, planar slabs, long vectors), i.e.
// it replaces the same reference code:
//   kernels/sparse_gemv.py:50-83    splitk_sparse_gemv_kernel   (mask + gathered GEMV)
//   kernels/sparse_gemv.py:152-194  qkv_kernel                  (three thresholds over one fused weight)
//   gpt-fast/model.py:158-161,258-259,289-291                   (residual adds, RMSNorm, silu(gate) * up: fused producers)
//
// Why a second kernel: on MI355X a 7B-class launch streams for 3-14 us, and the phase stamps of the general kernel
// showed 2.3 us between kernel entry and "row list ready" of which only ~0.8 us is memory latency — the rest is
// INSTRUCTION ISSUE: 16 waves share 4 SIMDs, so every instruction of the per-wave prologue costs four issue slots, and
// the general kernel executes ~800 of them (runtime geometry: integer division of the block index, segment lookup,
// clamps for ragged vectors, 64-bit address arithmetic, five dependent groups of scalar kernel-argument loads that
// each miss the scalar cache).  Here:
//   * the arguments the first loads need are separate scalar kernel parameters, preloaded into SGPRs by the command
//     processor (-mllvm -amdgpu-kernarg-preload-count): the activation loads leave at the first instruction;
//   * everything else arrives with ONE batch of scalar loads that overlaps the activation loads;
//   * the grid is (tiles, slices): no division; vectors are whole chunks and tiles whole: no clamps;
//   * 32-bit element offsets against uniform bases;
//   * phase stamps are a template flag (compiled out of the production instantiations).
#pragma once
#include "teal_gemv_fast_decl.h"
#include "teal_gemv_kernel.h"

namespace teal {

// ONE batch of scalar loads for every remaining kernel argument, placed right after the activation loads have been
// issued (an "s" input forces the value into an SGPR at this point of the program)
#define TEAL_FAST_ARGS_BATCH(a)                                                                                         \
    asm volatile("" ::"s"((a).w0), "s"((a).w1), "s"((a).y), "s"((a).ws), "s"((a).mask_out), "s"((a).resid_out),          \
                 "s"((a).phase), "s"((a).ticket), "s"((a).ld0), "s"((a).ld1), "s"((a).seg_tile1), "s"((a).seg_tile2), "s"((a).tau0),       \
                 "s"((a).tau1), "s"((a).tau2), "s"((a).mask_tau), "s"((a).ws_stride), "s"((a).att_hd), "s"((a).att_ns),  \
                 "s"((a).cap), "s"((a).w1_tile), "s"((a).scale0), "s"((a).scale1), "s"((a).sum32))


// LPR lanes x 16 B = BN columns per tile; KR = register-cached 64-element chunks per wave; U = 16-byte loads a lane has
// in flight per batch (two batches in the software pipeline).
//   MODE 0 plain x; 1 residual + slabs -> RMSNorm (in0 residual, in1 slabs, in2 norm weight); 2 x = silu(gate) * up with
//   gate|up contiguous [2Z] at in0 (gpt-fast/model.py:258-259, the roundings of the unfused sequence); 3 x + producer
//   masks (in0 x, in1 masks); 4 split-KV attention partials (in0).  Element-wise modes (0, 2, 3, 4) cache the rounds of
//   the workgroup's own slice only: register k <-> chunk slice + split * (wave + 16 k).
//   EXACT (MODE 1): Z == 1024 * KR, every cached chunk exists — no clamps, no guards.
template <bool BF16, int MODE, bool PAIR, int LPR, int KR, bool EXACT, bool PHASE, int U = 4, bool W8 = false, bool ROPE = false>
__global__ __launch_bounds__(1024) void gemv_fast_kernel(const void* in0, const void* in1, const void* in2,
                                                         const int* row_index, const int Z, const int nslabs,
                                                         const float eps, const int split_p, const FastArgs a) {
    static_assert(!(W8 && PAIR), "int8 gate|up runs unpaired (two images of 128-byte row segments)");
    static_assert(!ROPE || (MODE == 1 && !PAIR && !W8), "the RoPE / KV-append epilogue belongs to the fused wqkv projection");
    constexpr int WAVES = 16;
    constexpr int RPW = 64 / LPR;
    // a lane owns 8 columns: 16 bytes of 16-bit weights, 8 bytes of int8 weights (the same lane <-> row mapping, hence the
    // same summation order, as sparse_gemv_kernel's int8 form: bit-identical results).  16 int8 columns per lane (16-byte
    // loads, 8 lanes per 128-byte row segment) measured SLOWER: 669 against 706 tok/s on Llama-2-7B int8 @ 50 %.
    constexpr int CPL = 8;
    constexpr int BN = LPR * CPL;
    constexpr int WB = W8 ? 1 : 2;  // bytes per weight
    using wvec = std::conditional_t<W8, u32x2, u32x4>;
    unsigned long long t_entry = 0;
    if constexpr (PHASE) t_entry = wall_clock64();
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // split = gridDim.y, handed over as a PRELOADED scalar argument: gridDim lives in the dispatch packet / hidden kernel
    // arguments, i.e. behind a scalar load — and the element-wise producers need it for the address of their first activation
    // load (round 4: the ISA of the down / wo launches opened with s_load gridDim.y; s_waitcnt lgkmcnt(0) — the 0.4-0.5 us
    // "kernarg" phase of those launches)
    const int tile = blockIdx.x, slice = blockIdx.y, split = split_p;
    const int nch = Z >> 6;
    const uint32_t bid = blockIdx.y * gridDim.x + blockIdx.x;
    auto stamp = [&](const int i) {
        if constexpr (PHASE) { if (a.phase && tid == 0) a.phase[(size_t)bid * kPhaseRow + i] = wall_clock64(); }
    };

    // ---- producer: every load is issued before anything waits -----------------------------------------------
    const uint16_t* __restrict__ x16 = reinterpret_cast<const uint16_t*>(in0);
    uint32_t xr[KR];
    bool own[KR];    // chunk exists and its rows belong to this workgroup (wave-uniform)
    int cidx[KR];
    float sumsq_part = 0.0f;
    int rope_p = 0;            // ROPE: clamped position of the token
    uint32_t rope_cs = 0u;     // ROPE: (cos | sin << 16) of this thread's column pair
    float rv[MODE == 1 ? KR : 1];
    uint32_t wb[MODE == 1 ? KR : 1];
    unsigned long long mk[MODE == 3 ? KR : 1];
    {
        // Which chunks (64 activations) a workgroup streams — balanced over the slices to within one chunk (round 4; whole
        // rounds of 16 chunks per slice left Llama-2-7B's down projection, 172 chunks over 4 slices, at 48 / 48 / 44 / 32 and its
        // last workgroups 1.8 us behind the first: -0.2 % per token at 7B, -0.8 % at Llama-3-8B, -0.7 % at 70B widths):
        //   element-wise producers: chunk c belongs to slice c mod split, inside the slice to wave (c div split) mod 16;
        //   MODE 1 (every workgroup caches the whole vector, wave w holds chunks w + 16 k): chunk (w, k) belongs to slice
        //   (k + w) mod split — the slices' extra rounds rotate over the waves.
        // sparse_gemv_kernel (teal_gemv_kernel.h: chunk_of / kmod) uses the same rule: bit-identical outputs.
        // (own[] is filled in right before the compaction)
#pragma unroll
        for (int k = 0; k < KR; ++k) cidx[k] = MODE == 1 ? wave + WAVES * k : slice + split * (wave + WAVES * k);
    }
    if constexpr (MODE == 1) {
        const uint16_t* resid = x16;
        if (row_index) resid += (size_t)row_index[0] * (size_t)Z;  // embedding row of the current token
        const uint16_t* nw = reinterpret_cast<const uint16_t*>(in2);
        const float* slabs = reinterpret_cast<const float*>(in1);
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        uint32_t rb[KR];
        f32x4 v0[KR], v1[KR];
        // bit 8 of the (preloaded) slab count: ONE planar fp32 vector [Z] — the sum a TEAL_OUT_SLAB_SUM launch left (tensor
        // parallelism: what the ranks all-reduce) — instead of interleaved slabs [Z][(ns + 3) & ~3]
        const bool planar1 = (nslabs & 0x100) != 0;
        const int ns = nslabs & 0xFF;
        const uint32_t stride = (uint32_t)(ns + 3) & ~3u;
        uint32_t mel[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            mel[k] = (uint32_t)(EXACT ? cidx[k] : min(cidx[k], nch - 1)) * 64u + lane;
            rb[k] = resid[mel[k]];
            wb[k] = nw[mel[k]];
        }
        if (planar1) {
#pragma unroll
            for (int k = 0; k < KR; ++k) v0[k] = f32x4{slabs[mel[k]], 0.0f, 0.0f, 0.0f};
        } else if (ns > 0) {
#pragma unroll
            for (int k = 0; k < KR; ++k) v0[k] = *reinterpret_cast<const f32x4*>(slabs + mel[k] * stride);
        }
        if (ns > 4) {
#pragma unroll
            for (int k = 0; k < KR; ++k) v1[k] = *reinterpret_cast<const f32x4*>(slabs + mel[k] * stride + 4);
        }
        TEAL_FAST_ARGS_BATCH(a);
        if constexpr (ROPE) {
            // position and the (cos, sin) pair of this thread's output column: requested now, used in the epilogue.  Issued by
            // EVERY thread (unconditional: the compiler's vmcnt bookkeeping stays exact), the same 4 bytes per column pair
            asm volatile("" ::"s"(a.rope), "s"(a.rope_pos), "s"(a.kc), "s"(a.vc), "s"(a.rope_hd), "s"(a.rope_dim), "s"(a.rope_kv),
                         "s"(a.rope_max_seq));
            rope_p = min(max(a.rope_pos[0], 0), a.rope_max_seq - 1);
            const uint32_t cc = (uint32_t)tile * BN + (uint32_t)(tid & (BN - 1));
            const uint32_t jj = (cc & (uint32_t)(a.rope_hd - 1)) >> 1;  // head_dim is 64 or 128
            rope_cs = *reinterpret_cast<const uint32_t*>(a.rope + ((size_t)rope_p * (size_t)(a.rope_hd >> 1) + jj) * 2);
        }
        stamp(1);
        // slab order 0, 1, 2, ... (the order of the ordered reduce launch); adding an absent slab as 0.0f is exact
        float ysum[KR];
        if (ns == 4) {
#pragma unroll
            for (int k = 0; k < KR; ++k) ysum[k] = (((0.0f + v0[k][0]) + v0[k][1]) + v0[k][2]) + v0[k][3];
        } else if (ns > 0) {
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                float sacc = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) sacc += (j < ns) ? v0[k][j] : 0.0f;
                if (ns > 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) sacc += (4 + j < ns) ? v1[k][j] : 0.0f;
                }
                ysum[k] = sacc;
            }
        }
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            float r = bits_to_float(rb[k], BF16);
            if (ns > 0) {
                const float yv = bits_to_float(float_to_bits<BF16>(ysum[k]), BF16);
                r = bits_to_float(float_to_bits<BF16>(r + yv), BF16);
            }
            if (!EXACT) r = (cidx[k] < nch) ? r : 0.0f;
            rv[k] = r;
            sumsq_part = fmaf(r, r, sumsq_part);  // explicit: see the contract(off) note at the top of teal_gemv_kernel.h
        }
        stamp(8);
        float* sumsq = reinterpret_cast<float*>(smem);
        const float ssw = wave_sum_f(sumsq_part);
        if (lane == 0) sumsq[wave] = ssw;
        __syncthreads();
        stamp(9);
        float tot = (lane < WAVES) ? sumsq[lane] : 0.0f;
        tot = wave_sum_f(tot);
        const float rstd = rsqrtf(tot / (float)Z + eps);
        uint16_t* rout = reinterpret_cast<uint16_t*>(a.resid_out);
        const bool writer = rout && blockIdx.x == 0 && blockIdx.y == 0;  // (not via bid: gridDim.x is a scalar load)
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const float xn = bits_to_float(float_to_bits<BF16>(rv[k] * rstd), BF16);
            xr[k] = float_to_bits<BF16>(xn * bits_to_float(wb[k], BF16));
            if (writer && (EXACT || cidx[k] < nch)) rout[(uint32_t)cidx[k] * 64u + lane] = float_to_bits<BF16>(rv[k]);
        }
    } else if constexpr (MODE == 4) {
        // attention output merged from the split-KV partials {max, sum, o[hd]} per (head, split): see the merge
        // producer of sparse_gemv_kernel; one lane per (cached chunk, split) fetches {max, sum}
        // head_dim (64 / 128) and the partials per head arrive in the preloaded `nslabs` slot as att_ns | log2(head_dim) << 8:
        // nothing the merge's loads need waits for the argument batch, and head_dim divides by shifts
        const float* att = reinterpret_cast<const float*>(in0);
        const int hsh = nslabs >> 8, att_ns = nslabs & 0xFF;
        const int hd = 1 << hsh, hs = hd + 2;
        auto merge = [&](auto ns_tag) {
            constexpr int NS = decltype(ns_tag)::value;
            static_assert(KR * NS <= 64, "one lane per (chunk, split)");
            const int kk = min(lane / NS, KR - 1), qq = lane % NS;
            const int ck = min(slice + split * (wave + WAVES * kk), nch - 1);
            const float2 st = *reinterpret_cast<const float2*>(att + ((size_t)((ck << 6) >> hsh) * NS + qq) * hs);
            float ov[KR][NS];
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int m = (min(cidx[k], nch - 1) << 6) + lane;
                const int h = m >> hsh, d = m & (hd - 1);
                const float* b = att + (size_t)h * NS * hs + 2 + d;
#pragma unroll
                for (int q = 0; q < NS; ++q) ov[k][q] = b[q * hs];
            }
            TEAL_FAST_ARGS_BATCH(a);
            stamp(1);
#define TEAL_DPPF(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
            float M = fmaxf(st.x, TEAL_DPPF(st.x, 0xB1));
            M = fmaxf(M, TEAL_DPPF(M, 0x4E));
            if constexpr (NS == 8) M = fmaxf(M, TEAL_DPPF(M, 0x141));
            const float f = st.y > 0.0f ? expf(st.x - M) : 0.0f;
            float Ls = st.y * f;
            Ls += TEAL_DPPF(Ls, 0xB1);
            Ls += TEAL_DPPF(Ls, 0x4E);
            if constexpr (NS == 8) Ls += TEAL_DPPF(Ls, 0x141);
#undef TEAL_DPPF
            const int cw = __float_as_int(f / Ls);
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                float Os = 0.0f;
#pragma unroll
                for (int q = 0; q < NS; ++q) Os = fmaf(ov[k][q], __int_as_float(__builtin_amdgcn_readlane(cw, NS * k + q)), Os);
                xr[k] = float_to_bits<BF16>(Os);
            }
        };
        if constexpr (KR * 8 <= 64) {
            if (att_ns == 8) merge(std::integral_constant<int, 8>{});
            else merge(std::integral_constant<int, 4>{});
        } else {
            merge(std::integral_constant<int, 4>{});
        }
    } else if constexpr (MODE == 2) {
        uint32_t gb[KR], ub[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) {  // all gate / up loads first, then the activation maths
            const uint32_t m = (uint32_t)min(cidx[k], nch - 1) * 64u + lane;
            gb[k] = x16[m];
            ub[k] = x16[(uint32_t)Z + m];
        }
        TEAL_FAST_ARGS_BATCH(a);
        stamp(1);
        if (a.gate_act) {  // the gate half went through silu in the gate | up launch's epilogue (act_seg0): once per column there,
                           // instead of ~30 instructions per element in the prologue of every one of this launch's workgroups
#pragma unroll
            for (int k = 0; k < KR; ++k) xr[k] = float_to_bits<BF16>(bits_to_float(gb[k], BF16) * bits_to_float(ub[k], BF16));
        } else {
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const float gt = bits_to_float(gb[k], BF16);
            const float sl = bits_to_float(float_to_bits<BF16>(gt / (1.0f + expf(-gt))), BF16);
            xr[k] = float_to_bits<BF16>(sl * bits_to_float(ub[k], BF16));
        }
        }
    } else {
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int c = min(cidx[k], nch - 1);
            xr[k] = x16[(uint32_t)c * 64u + lane];
            if constexpr (MODE == 3) mk[k] = reinterpret_cast<const unsigned long long*>(in1)[c];
        }
        TEAL_FAST_ARGS_BATCH(a);
        stamp(1);
    }
    stamp(2);
    // int8: the per-column scale of the epilogue, requested now (a dependent epilogue load would cost a cold round trip)
    uint32_t scb = 0u;
    if constexpr (W8) {
        if (tid < BN) {
            const bool sec = tile >= a.w1_tile;
            scb = (sec ? a.scale1 : a.scale0)[(uint32_t)(sec ? tile - a.w1_tile : tile) * BN + tid];
        }
    }

    // ---- mask + wave-local compaction: (row:16 | x:16) pairs of the wave's own chunks, ascending ---------------
    int s = 0;
    if (tile >= a.seg_tile1) s = 1;
    if (tile >= a.seg_tile2) s = 2;
    const float tau_s = s == 0 ? a.tau0 : (s == 1 ? a.tau1 : a.tau2);
    const float tau = PAIR ? fminf(a.tau0, a.tau1) : tau_s;  // PAIR: union of the two keep sets
    {
        int kmod = (MODE == 1 && split > 1) ? wave % split : 0;
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            own[k] = (EXACT || cidx[k] < nch) && (MODE != 1 || kmod == slice);
            kmod = (kmod + 1 == split) ? 0 : kmod + 1;
        }
    }
    uint32_t* list = reinterpret_cast<uint32_t*>(smem + 64) + (size_t)wave * a.cap;
    float* red = reinterpret_cast<float*>(smem + 64 + (size_t)WAVES * a.cap * 4);
    int nloc = 0;
    bool prio_set = false;
    // ---- stream set-up first: the loads of the first batch leave as soon as the wave's FIRST chunk is compacted, while
    //      the remaining chunks are still being balloted (the launch is bound by HBM from the first request on, so every
    //      100 ns the pipeline starts earlier is 100 ns off the launch) -------------------------------------------------
    const int g = lane / LPR, cl = lane % LPR;
    // two weight images without pairing (gate | up as two threshold segments, each tile streams ONE matrix): tiles from
    // w1_tile on belong to the second image
    const bool second = !PAIR && tile >= a.w1_tile;
    const uint32_t col = (uint32_t)(second ? tile - a.w1_tile : tile) * BN + cl * CPL;
    const char* wp = reinterpret_cast<const char*>(second ? a.w1 : a.w0) + (size_t)col * WB;
    const uint32_t ldb = (uint32_t)(second ? a.ld1 : a.ld0) * (uint32_t)WB;
    const char* wp2 = PAIR ? reinterpret_cast<const char*>(a.w1) + (size_t)col * WB : nullptr;
    const uint32_t ldb2 = PAIR ? (uint32_t)a.ld1 * (uint32_t)WB : 0u;
    float xs = 0.0f;  // int8: sum of the activations multiplied into acc (bias correction of the epilogue, see fma8)
    const float tau_g = a.tau0, tau_u = a.tau1;
    const bool two_tau = PAIR && (tau_g != tau_u);
    float acc[CPL], acc2[PAIR ? 8 : 1];
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < (PAIR ? 8 : 1); ++j) acc2[j] = 0.0f;
    {
        constexpr int STEP = U * RPW;
        auto full = [&](const int e) { return e + STEP <= nloc; };
        auto issue = [&](wvec (&w)[U], wvec (&w2)[PAIR ? U : 1], float (&xv)[U], const int e0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t ent = list[e0 + u * RPW + g];
                xv[u] = bits_to_float(ent & 0xFFFFu, BF16);
                w[u] = __builtin_nontemporal_load(reinterpret_cast<const wvec*>(wp + (size_t)(ent >> 16) * ldb));
                if constexpr (PAIR)
                    w2[u] = __builtin_nontemporal_load(reinterpret_cast<const wvec*>(wp2 + (size_t)(ent >> 16) * ldb2));
            }
        };
        auto consume = [&](wvec (&w)[U], wvec (&w2)[PAIR ? U : 1], float (&xv)[U]) {
            if (two_tau) {  // a row outside one of the two keep sets: zero weights, exactly a masked load
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
                if constexpr (W8) xs += xv[u];
                if constexpr (PAIR) fma8<BF16>(acc2, w2[u], xv[u]);
            }
        };
        wvec wa[U], wbb[U], w2a[PAIR ? U : 1], w2b[PAIR ? U : 1];
        float xa[U], xb[U];
        bool early = false;
        // ---- mask + wave-local compaction ----------------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            if (own[k]) {
                unsigned long long mask;
                if constexpr (MODE == 3) {
                    mask = mk[k];
                } else {
                    const float v = bits_to_float(xr[k], BF16);
                    mask = __ballot(keep_rule(v, tau) || (v != v));  // NaN propagates like the reference's 0 * NaN
                }
                if ((mask >> lane) & 1ull) list[nloc + lane_rank(mask)] = (((uint32_t)cidx[k] * 64u + lane) << 16) | xr[k];
                nloc += __popcll(mask);
            }
            if (k == 0 && KR > 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (full(0)) {
                    issue(wa, w2a, xa, 0);
                    early = true;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        stamp(3);
        // Short lists (fewer than six batches per wave: the sparse launches of 4096-wide models): issue priority rising
        // with the wave index.  Measured, not derived: the arbiter serves the oldest wave first, and with only a few
        // batches per wave letting the youngest go first is worth 1.0 % of a Llama-2-7B token (0.7 % on Llama-3-8B),
        // while long lists lose 0.3-0.7 % (every row kept, or 8192-wide models) and keep the default.  Timing only.
        prio_set = nloc < 6 * 4 * RPW;
        if (prio_set) {
            switch (wave >> 2) {  // (oldest first made explicit: -1.2 %; interleaved, wave & 3: -0.8 % against this ramp)
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                case 3: __builtin_amdgcn_s_setprio(3); break;
                default: break;
            }
        }
        int eb = 0;
        bool fa = early || full(eb);
        bool first_done = false;
        if (fa && !early) issue(wa, w2a, xa, eb);
        while (fa) {
            int ebn = eb + STEP;
            const bool fb = full(ebn);
            if (fb) issue(wbb, w2b, xb, ebn);
            consume(wa, w2a, xa);
            if constexpr (PHASE) { if (!first_done) { first_done = true; stamp(4); } }
            eb = ebn;
            if (!fb) break;
            ebn = eb + STEP;
            fa = full(ebn);
            if (fa) issue(wa, w2a, xa, ebn);
            consume(wbb, w2b, xb);
            eb = ebn;
        }
        if (eb < nloc) {  // tail: clamp the entry index, zero the weights of clamped lanes
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = eb + u * RPW + g;
                const bool ok = e < nloc;
                const uint32_t ent = list[ok ? e : nloc - 1];
                xa[u] = ok ? bits_to_float(ent & 0xFFFFu, BF16) : 0.0f;
                wvec t = __builtin_nontemporal_load(reinterpret_cast<const wvec*>(wp + (size_t)(ent >> 16) * ldb));
                if (!ok) t = wvec(0u);  // (int8: the zero activation alone drops the row and keeps it out of the bias sum)
                wa[u] = t;
                if constexpr (PAIR) {
                    wvec t2 = __builtin_nontemporal_load(reinterpret_cast<const wvec*>(wp2 + (size_t)(ent >> 16) * ldb2));
                    if (!ok) t2 = wvec(0u);
                    w2a[u] = t2;
                }
            }
            consume(wa, w2a, xa);
        }
    }
    stamp(5);
    if (prio_set) __builtin_amdgcn_s_setprio(0);
    if constexpr (PHASE) { if (a.phase && lane == 0) a.phase[(size_t)bid * kPhaseRow + 16 + wave] = wall_clock64(); }

    // ---- reduce: row groups of the wave (shuffles), then waves in fixed order through LDS -------------------
    // lane ^ 8 as a DPP row rotate, lane ^ 16 as a ds_swizzle swap, lane ^ 32 a shuffle (round 2: +0.7 % tokens/s over shuffles for
    // every step; same pairs, same order: bit-identical)
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        if constexpr (LPR <= 8) acc[j] = xor_add<8>(acc[j]);
        acc[j] = xor_add<32>(xor_add<16>(acc[j]));
    }
    if constexpr (PAIR) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (LPR <= 8) acc2[j] = xor_add<8>(acc2[j]);
            acc2[j] = xor_add<32>(xor_add<16>(acc2[j]));
        }
    }
    if (lane < LPR) {
        float* r = red + wave * BN + lane * CPL;
#pragma unroll
        for (int j = 0; j < CPL; ++j) r[j] = acc[j];
        if constexpr (PAIR) {
            float* r2 = red + (WAVES + wave) * BN + lane * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) r2[j] = acc2[j];
        }
    }
    float* xsw = red + (PAIR ? 2 : 1) * WAVES * BN;  // int8: [WAVES] per-wave activation sums
    if constexpr (W8) {
        if constexpr (LPR <= 8) xs = xor_add<8>(xs);
        xs = xor_add<32>(xor_add<16>(xs));
        if (lane == 0) xsw[wave] = xs;
    }
    __syncthreads();
    stamp(6);
    if (tid < BN) {  // whole waves: BN is a multiple of 64
        const uint32_t c = (uint32_t)tile * BN + tid;
        float gs = 0.0f, us = 0.0f;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) {
            gs += red[wv * BN + tid];
            if constexpr (PAIR) us += red[(WAVES + wv) * BN + tid];
        }
        if constexpr (W8) {
            // sum q * x = sum (q + 1152) * x - 1152 * sum x (fma8 on bytes), then the per-column scale in fp32 before the one
            // rounding (gpt-fast/quantize.py:354 scales the reduced product)
            float bias = 0.0f;
#pragma unroll
            for (int wv = 0; wv < WAVES; ++wv) bias += xsw[wv];
            bias *= kInt8Bias;
            gs = (gs - bias) * bits_to_float(scb, BF16);
        }
        if constexpr (PAIR) {
            // h = silu(gate) * up with the roundings of the unfused sequence (gpt-fast/model.py:258-259), and the keep
            // masks of h against the down projection's threshold for a MODE 3 consumer
            const float g16 = bits_to_float(float_to_bits<BF16>(gs), BF16);
            const float u16 = bits_to_float(float_to_bits<BF16>(us), BF16);
            const float sl = bits_to_float(float_to_bits<BF16>(g16 / (1.0f + expf(-g16))), BF16);
            const uint32_t hb = float_to_bits<BF16>(sl * u16);
            reinterpret_cast<uint16_t*>(a.y)[c] = (uint16_t)hb;
            const float hv = bits_to_float(hb, BF16);
            const unsigned long long mko = __ballot(keep_rule(hv, a.mask_tau) || (hv != hv));
            if (a.mask_out && lane == 0) a.mask_out[c >> 6] = mko;
        } else if (a.ws_stride == 0) {
            if constexpr (ROPE) {
                // gpt-fast/model.py:170-178: q and the new k row are rotated (pairs of adjacent columns: lane ^ 1 holds the
                // partner), k and v go straight into their cache rows; the roundings are those of the unfused sequence
                // (projection rounded, rotation in fp32, rounded again — what the attention launch did with the same helpers)
                const uint16_t b16 = float_to_bits<BF16>(gs);
                const float f = bits_to_float(b16, BF16);
                const float pr = __shfl_xor(f, 1);
                const float cs = bits_to_float(rope_cs & 0xFFFFu, BF16), sn = bits_to_float(rope_cs >> 16, BF16);
                const uint16_t rb = float_to_bits<BF16>((tid & 1) ? rope_odd(pr, f, cs, sn) : rope_even(f, pr, cs, sn));
                const uint32_t dimq = (uint32_t)a.rope_dim, kvw = (uint32_t)a.rope_kv, hdm = (uint32_t)a.rope_hd;
                if (c < dimq) {
                    reinterpret_cast<uint16_t*>(a.y)[c] = rb;
                } else {
                    const bool isk = c < dimq + kvw;
                    const uint32_t cc = c - dimq - (isk ? 0u : kvw);
                    const uint32_t kvh = cc / hdm, d = cc & (hdm - 1u);
                    (isk ? a.kc : a.vc)[((size_t)kvh * (size_t)a.rope_max_seq + (size_t)rope_p) * hdm + d] = isk ? rb : b16;
                }
            } else if (a.act0 && s == 0) {  // the gate tiles of gate | up: activation applied here (model.py:258)
                reinterpret_cast<uint16_t*>(a.y)[c] = silu_bits<BF16>(gs);
            } else if (a.sum32) {           // TEAL_OUT_SLAB_SUM without split-K: the unrounded fp32 sum
                reinterpret_cast<float*>(a.y)[c] = gs;
            } else {
                reinterpret_cast<uint16_t*>(a.y)[c] = float_to_bits<BF16>(gs);
            }
        } else if (!a.ticket) {
            a.ws[c * (uint32_t)a.ws_stride + slice] = gs;
        } else {
            // publish the partial write-through (agent-scope relaxed atomic store = global_store ... sc1)
            __hip_atomic_store(&a.ws[c * (uint32_t)a.ws_stride + slice], gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (!PAIR) {
        if (a.ticket) {
            // Split-K with ONE launch (replaces the reference's memset + fp16 atomics, kernels/sparse_gemv.py:8-12,83, and
            // the ordered reduce launch): every slice of a tile publishes its fp32 partial write-through, drains, and takes
            // a ticket; the last to arrive sums the `split` partials IN SLICE ORDER (bit-identical to the reduce launch, no
            // matter who arrives last), rounds once, stores y and re-arms the counter.
            float* tflag = reinterpret_cast<float*>(smem);  // the RMSNorm scratch is free by now
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
            __syncthreads();
            if (tid == 0) {
                const unsigned t = __hip_atomic_fetch_add(&a.ticket[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tflag[0] = (t == (unsigned)split - 1u) ? 1.0f : 0.0f;
            }
            __syncthreads();
            if (tflag[0] != 0.0f && tid < BN) {
                const uint32_t c = (uint32_t)tile * BN + tid;
                float sum = 0.0f;
                for (int sl = 0; sl < split; ++sl)
                    sum += __hip_atomic_load(&a.ws[c * (uint32_t)a.ws_stride + sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.sum32) reinterpret_cast<float*>(a.y)[c] = sum;  // TEAL_OUT_SLAB_SUM: slice-order fp32 sum, rounded by the consumer
                else reinterpret_cast<uint16_t*>(a.y)[c] = (a.act0 && s == 0) ? silu_bits<BF16>(sum) : float_to_bits<BF16>(sum);
                if (tid == 0) __hip_atomic_store(&a.ticket[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    stamp(7);
    if constexpr (PHASE) {
        if (a.phase && tid == 0) {
            unsigned xcc = 0, hwid = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            a.phase[(size_t)bid * kPhaseRow] = t_entry;
            a.phase[(size_t)bid * kPhaseRow + 12] = ((unsigned long long)hwid << 32) | xcc;
            a.phase[(size_t)bid * kPhaseRow + 13] = ((unsigned long long)WAVES << 32) | (gridDim.x * gridDim.y);
        }
    }
}


template <bool BF16, int MODE, bool PAIR, int LPR, int KR, bool EXACT, bool W8 = false>
hipError_t launch_fast_e(const FastLaunch& f, hipStream_t st) {
    const dim3 grid(f.ntiles, f.split), block(1024);
#ifdef TEAL_DIAGNOSTICS
    if constexpr (!W8) {
        if (f.a.phase) {  // stamped instantiations (libteal_hip_diag.so: teal_set_phase_buffer), both activation dtypes
            if constexpr (MODE == 1 && !PAIR) {
                if (f.a.rope) {  // the stamped form of the RoPE / KV-append epilogue: a rope request never runs unrotated
                    hipLaunchKernelGGL((gemv_fast_kernel<BF16, MODE, PAIR, LPR, KR, EXACT, true, 4, false, true>), grid, block, f.lds, st,
                                       f.in0, f.in1, f.in2, f.row_index, f.Z, f.nslabs, f.eps, f.split, f.a);
                    return hipGetLastError();
                }
            }
            hipLaunchKernelGGL((gemv_fast_kernel<BF16, MODE, PAIR, LPR, KR, EXACT, true>), grid, block, f.lds, st, f.in0, f.in1,
                               f.in2, f.row_index, f.Z, f.nslabs, f.eps, f.split, f.a);
            return hipGetLastError();
        }
    }
#endif
    if constexpr (MODE == 1 && !PAIR && !W8) {
        if (f.a.rope) {  // fused wqkv projection, split == 1: RoPE + KV-cache append in the epilogue
            hipLaunchKernelGGL((gemv_fast_kernel<BF16, MODE, PAIR, LPR, KR, EXACT, false, 4, W8, true>), grid, block, f.lds, st, f.in0, f.in1,
                               f.in2, f.row_index, f.Z, f.nslabs, f.eps, f.split, f.a);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((gemv_fast_kernel<BF16, MODE, PAIR, LPR, KR, EXACT, false, 4, W8>), grid, block, f.lds, st, f.in0, f.in1, f.in2,
                       f.row_index, f.Z, f.nslabs, f.eps, f.split, f.a);
    return hipGetLastError();
}

template <bool BF16, int MODE, bool PAIR, int LPR, int KR, bool W8 = false>
hipError_t launch_fast_k(const FastLaunch& f, hipStream_t st) {
    if constexpr (MODE == 1) {
        if (f.Z == 1024 * KR) return launch_fast_e<BF16, MODE, PAIR, LPR, KR, true, W8>(f, st);
    }
    return launch_fast_e<BF16, MODE, PAIR, LPR, KR, false, W8>(f, st);
}

template <bool BF16, int MODE, bool PAIR, int LPR, bool W8 = false>
hipError_t launch_fast_r(const FastLaunch& f, hipStream_t st) {
    if constexpr (MODE == 1) {
        switch (f.kr) {
            case 4: return launch_fast_k<BF16, MODE, PAIR, LPR, 4, W8>(f, st);
            case 8: return launch_fast_k<BF16, MODE, PAIR, LPR, 8, W8>(f, st);
            case 16: return launch_fast_k<BF16, MODE, PAIR, LPR, 16, W8>(f, st);
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (f.kr) {
            case 1: return launch_fast_k<BF16, MODE, PAIR, LPR, 1, W8>(f, st);
            case 4: return launch_fast_k<BF16, MODE, PAIR, LPR, 4, W8>(f, st);
            case 8: return launch_fast_k<BF16, MODE, PAIR, LPR, 8, W8>(f, st);
            default: return hipErrorInvalidValue;
        }
    }
}

template <bool BF16, int LPR>
hipError_t launch_fast_m(const FastLaunch& f, hipStream_t st) {
    if (f.pair) return f.mode == 1 ? launch_fast_r<BF16, 1, true, LPR>(f, st) : hipErrorInvalidValue;
    switch (f.mode) {
        case 0: return launch_fast_r<BF16, 0, false, LPR>(f, st);
        case 1: return launch_fast_r<BF16, 1, false, LPR>(f, st);
        case 2: return launch_fast_r<BF16, 2, false, LPR>(f, st);
        case 3: return launch_fast_r<BF16, 3, false, LPR>(f, st);
        case 4: return launch_fast_r<BF16, 4, false, LPR>(f, st);
        default: return hipErrorInvalidValue;
    }
}

template <bool BF16>
hipError_t launch_fast_q(const FastLaunch& f, hipStream_t st) {
    switch (f.lpr) {
        case 8: return launch_fast_m<BF16, 8>(f, st);
        case 16: return launch_fast_m<BF16, 16>(f, st);
        default: return hipErrorInvalidValue;
    }
}

// int8 weight images: 16 lanes x 8 bytes = 128-column tiles (128-byte row segments), never paired
template <bool BF16>
hipError_t launch_fast_w8_q(const FastLaunch& f, hipStream_t st) {
    if (f.pair || f.lpr != 16) return hipErrorInvalidValue;
    switch (f.mode) {
        case 0: return launch_fast_r<BF16, 0, false, 16, true>(f, st);
        case 1: return launch_fast_r<BF16, 1, false, 16, true>(f, st);
        case 2: return launch_fast_r<BF16, 2, false, 16, true>(f, st);
        case 3: return launch_fast_r<BF16, 3, false, 16, true>(f, st);
        case 4: return launch_fast_r<BF16, 4, false, 16, true>(f, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace teal
