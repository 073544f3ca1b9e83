// teal_prefill.hip — the DENSE prompt pass for short prompts (T <= 16 tokens), hand-fused for gfx950 / CDNA4, wave64.
//
// The reference's prefill is dense by construction: its ops fall back to torch.matmul when the sequence is longer than one
// token (kernels/sparse_gemv.py:271,298), and the rest of the prompt pass is the stock gpt-fast model (gpt-fast/model.py:
// 107-121 forward, 158-161 block, 170-186 attention, 258-259 feed-forward, 289-291 RMSNorm).  The reference's tokens/sec
// counts that pass (gpt-fast/generate.py:458,487-496), and on MI355X an op-by-op prompt pass of a 6-token prompt costs as much
// as nine decode steps (~400 launches of a few microseconds each; profiles/r05_generate_breakdown_before.txt: 20 ms eager,
// 10 ms replayed from a hipGraph).  Here one layer is seven launches over the SAME weight images the decode step streams:
//
//   gemm(wqkv) [RMSNorm while staging] -> attention (RoPE, cache rows 0..T-1, causal softmax) -> gemm(wo) -> resid ->
//   gemm(w1 | w3) [RMSNorm while staging] -> gemm(w2) [silu * up while staging] -> resid
//
// Every hand-over between launches is TRANSPOSED: [feature][KR], KR = 8 for T <= 8 and 16 for 9 <= T <= 16 (round 6: prompts of
// 9-16 tokens cost 2.1-2.2x the 8-token pass through the module path, profiles/r06_prefill_vs_prompt_length.txt) — the tokens of a
// feature are one or two 16-byte words (16-bit activations) or KR / 4 of them (fp32 split-K slabs) — so that "row m of every
// token" is contiguous: a GEMM workgroup
// stages its slice's rows in LDS once, and a wave then fetches a row with one broadcast read.  The GEMM is bound by HBM like the GEMV (every weight
// byte is read once for all tokens: 2 * T flops per byte, far from MFMA territory at T <= 8, and the reduction dimension is
// the strided one in the W^T image, which rules the matrix cores' operand layout out without a second copy of the weights).
// Rounding points are those of the module path's 16-bit tensors (projection outputs, RoPE, attention output, residual adds,
// RMSNorm twice, silu, product); sums are fp32.  Floating-point parity is against the module path, tolerance in the tests.
#include "teal_common.h"

#include <limits.h>

namespace teal {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kRowsMax = 16;  // most tokens a pass takes; a feature's row holds rows_for(T) of them
constexpr int kPrefillMaxSplit = 16;
constexpr int rows_for(const int T) { return T <= 8 ? 8 : 16; }

// sum of `split` slabs in slice order, rounded once to the activation dtype: what a projection's 16-bit output tensor holds
template <bool BF16, int KR>
__device__ __forceinline__ void rounded_row(const float* __restrict__ slabs, const int split, const size_t n_total, const size_t col,
                                            float (&out)[KR]) {
    constexpr int NW = KR / 4, UU = KR == 8 ? 4 : 2;  // 16-byte words per row; slices whose loads are in flight together
    f32x4 acc[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) acc[w] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < split; k0 += UU) {  // (clamped: a repeated slice is not added); slice order kept
        f32x4 v[UU][NW];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const float* p = slabs + ((size_t)min(k0 + u, split - 1) * n_total + col) * KR;
#pragma unroll
            for (int w = 0; w < NW; ++w) v[u][w] = *reinterpret_cast<const f32x4*>(p + 4 * w);
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            if (k0 + u < split) {
#pragma unroll
                for (int w = 0; w < NW; ++w) acc[w] += v[u][w];
            }
        }
    }
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int s = 0; s < 4; ++s) out[4 * w + s] = bits_to_float(float_to_bits<BF16>(acc[w][s]), BF16);
}

// a feature's KR tokens as KR / 8 16-byte words of 16-bit values
template <bool BF16, int KR>
__device__ __forceinline__ void store_row(uint16_t* __restrict__ p, const float (&v)[KR]) {
#pragma unroll
    for (int w = 0; w < KR / 8; ++w) {
        u32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j] = (uint32_t)float_to_bits<BF16>(v[8 * w + 2 * j]) | ((uint32_t)float_to_bits<BF16>(v[8 * w + 2 * j + 1]) << 16);
        *reinterpret_cast<u32x4*>(p + 8 * w) = r;
    }
}

template <bool BF16, int KR>
__device__ __forceinline__ void load_row(const uint16_t* __restrict__ p, float (&x)[KR]) {
#pragma unroll
    for (int w = 0; w < KR / 8; ++w) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p + 8 * w);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[8 * w + 2 * j] = bits_to_float(v[j] & 0xFFFFu, BF16); x[8 * w + 2 * j + 1] = bits_to_float(v[j] >> 16, BF16); }
    }
}

// 1 / rms per token from the per-workgroup sums of squares sumsq [nwg][KR] (lane = producing workgroup, nwg <= 64; the total in
// workgroup order)
template <int KR>
__device__ __forceinline__ void rstd_rows(const float* __restrict__ sumsq, const int nwg, const int lane, const int Z, const float eps,
                                          float (&rstd)[KR]) {
    f32x4 pw[KR / 4];
#pragma unroll
    for (int w = 0; w < KR / 4; ++w) pw[w] = lane < nwg ? *reinterpret_cast<const f32x4*>(sumsq + (size_t)lane * KR + 4 * w) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KR; ++s) rstd[s] = rsqrtf(wave_sum_f(pw[s >> 2][s & 3]) / (float)Z + eps);
}

// ------------------------------------------------------------------------------------------------
// slabs[slice][n][8] (fp32) = sum over the slice's rows m of W^T[m][n] * x[token][m], for up to 2 NP tokens.
//   xt     [Z][8] 16-bit: xt[m][s] = activation m of token s
//   w0/w1  one or two W^T images [Z][ld]: column tiles < tiles0 stream w0, the others w1 (gate | up in one launch)
//   grid (256-column tiles, row slices); rows in groups of 16 (one per wave): group q belongs to slice q mod split
// One WAVE streams one weight row at a time across the whole 256-column tile: 64 lanes x 4 columns = a 512-byte contiguous row
// segment per load, and the row's activations are WAVE-UNIFORM: the workgroup stages its slice's rows of xt in LDS once and a
// wave fetches a row's eight tokens with one broadcast LDS read (through scalar loads instead, the loop carried 21 scalar
// instructions per row — addresses, clamps, selects — and a full `s_waitcnt lgkmcnt(0)` per batch: 3.8 TB/s).  Sixteen rows
// per wave are in flight (two batches of 8 x 8 bytes per lane: 128 KB per CU, what the GEMV keeps in flight); a lane's sums never
// leave the lane until the epilogue (no butterflies), where the 16 waves are added in fixed order through LDS, one token pair
// per round.  Accumulation over pairs of tokens (packed v_pk_fma_f32 until round 6; plain fused multiply-adds since the library
// is built without packed fp32, profiles/r06_concurrent_packed_fp32.txt: the 6-token pass 4.09 -> 4.25 ms).  First build (row groups inside a wave, the
// activations through vector loads, 8 rows in flight): 2.1-3.5 TB/s; profiles/r05_prefill_kernel_stats.txt.
// ------------------------------------------------------------------------------------------------
// What the staging loop builds a row's eight activations from (PROD): 0 = xt itself; 1 = RMSNorm of the residual rows ht with the
// per-workgroup sums of squares the resid launch left (every wave adds them itself) and the norm weight; 2 = silu(gate) * up from
// the slabs of the gate | up launch.  Folding these into the staging removes a launch each: every workgroup builds only the rows
// of its own slice (a 16th of the vector in the narrow projections), once.
struct PrefillProd {
    const uint16_t* xt;       // PROD 0: [Z][8];  PROD 1: the residual rows ht [Z][8]
    const float* sumsq;       // PROD 1: [nwg][8]
    const uint16_t* norm_w;   // PROD 1: [Z]
    const float* gu;          // PROD 2: slabs [gu_split][2 Z][8] of the gate | up launch
    float eps;
    int nwg, gu_split, T;
};

template <bool BF16, int NP, int PROD, int KR>
__global__ __launch_bounds__(1024) void prefill_gemm_kernel(const PrefillProd pr, const uint16_t* __restrict__ w0, const int ld0,
                                                            const uint16_t* __restrict__ w1, const int ld1, const int tiles0,
                                                            float* __restrict__ slabs, const int Z, const int n_total) {
    const uint16_t* __restrict__ xt = pr.xt;
    // KR = 16 (9-16 tokens): twice the accumulators per lane, so half the weight rows in flight per batch (the loop is bound by
    // its multiply-adds there, not by the loads) and half the row groups per staging phase (the same 32 KB of activations)
    constexpr int WAVES = 16, CPL = 4, BN = 256, U = KR == 8 ? 8 : 4, KW = KR / 8, PHASE_GROUPS = 1024 / KR;
    static_assert(NP <= KR / 2 && NP > (KR == 8 ? 0 : 4), "token pairs of this row width");
    extern __shared__ __align__(16) unsigned char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem);   // the slice's activation rows of the current phase: xs[(j * 16 + wave) * KW + word]
    float* red = reinterpret_cast<float*>(smem);  // epilogue (after a barrier): [WAVES][BN][2], one token pair per round
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, slice = blockIdx.y, split = gridDim.y;
    const bool second = tile >= tiles0;
    const uint32_t ld = (uint32_t)(second ? ld1 : ld0);
    const int ngroups = Z >> 4;
    const int nj = (ngroups - slice + split - 1) / split;  // row groups of this slice: q = slice + split * j, row = q * 16 + wave
    // this wave's rows: base + j * stride (elements)
    // (a UNIFORM row pointer indexed by the lane: the address is an SGPR base + one 32-bit lane offset, not a 64-bit add per load)
    const uint16_t* wrow = (second ? w1 : w0) + (size_t)(second ? tile - tiles0 : tile) * BN + (size_t)(slice * 16 + wave) * ld;
    const size_t stride = (size_t)split * 16u * ld;
    f32x2 acc[CPL][NP];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[c][p] = (f32x2){0.0f, 0.0f};
    auto consume = [&](const u32x2 w, const u32x4 (&xv)[KW]) {
        f32x2 xp[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) xp[p] = (f32x2){bits_to_float(xv[p >> 2][p & 3] & 0xFFFFu, BF16), bits_to_float(xv[p >> 2][p & 3] >> 16, BF16)};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float wl = bits_to_float(w[j] & 0xFFFFu, BF16), wh = bits_to_float(w[j] >> 16, BF16);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                acc[2 * j][p] = __builtin_elementwise_fma((f32x2){wl, wl}, xp[p], acc[2 * j][p]);
                acc[2 * j + 1][p] = __builtin_elementwise_fma((f32x2){wh, wh}, xp[p], acc[2 * j + 1][p]);
            }
        }
    };
    for (int jb = 0; jb < nj; jb += PHASE_GROUPS) {
        const int njp = min(PHASE_GROUPS, nj - jb);  // groups of this phase (one phase for every Llama-2-7B launch)
        const uint16_t* wph = wrow + (size_t)jb * stride;
        const int nfull = njp / (2 * U);  // whole pairs of batches
        // (staging the rows as fp32 — no conversion of the activations in the loop: 17 instead of 25 vector instructions per row —
        //  was measured too: no faster, gate | up 41.3 -> 43.6 us: the loop is not bound by its conversions)
        // (requesting the first batch of weight rows ahead of the staging — nothing in it depends on the activations — was
        //  measured: the 16 registers it keeps live through the producers push the kernel to 126-128 registers with spills, and
        //  the gate | up launch went 41 -> 48 us; profiles/r05_prefill_kernel_stats.txt)
        if (jb) __syncthreads();
        [[maybe_unused]] float rstd[KR];
        if constexpr (PROD == 1) rstd_rows<KR>(pr.sumsq, pr.nwg, lane, Z, pr.eps, rstd);
        for (int r = tid; r < njp * 16; r += 1024) {
            const uint32_t m = (uint32_t)(slice + split * (jb + (r >> 4))) * 16u + (uint32_t)(r & 15);
            [[maybe_unused]] uint16_t* dst = reinterpret_cast<uint16_t*>(xs + (size_t)r * KW);
            if constexpr (PROD == 0) {
#pragma unroll
                for (int w = 0; w < KW; ++w) xs[(size_t)r * KW + w] = *reinterpret_cast<const u32x4*>(xt + (size_t)m * KR + 8 * w);
            } else if constexpr (PROD == 1) {  // x = round(round(h * rstd) * w)  (gpt-fast/model.py:289-291)
                const float nw = bits_to_float(pr.norm_w[m], BF16);
                float x[KR];
                load_row<BF16, KR>(xt + (size_t)m * KR, x);
#pragma unroll
                for (int s_ = 0; s_ < KR; ++s_) {
                    const float xn = bits_to_float(float_to_bits<BF16>(x[s_] * rstd[s_]), BF16);
                    x[s_] = s_ < pr.T ? bits_to_float(float_to_bits<BF16>(xn * nw), BF16) : 0.0f;
                }
                store_row<BF16, KR>(dst, x);
            } else {  // x = round(round(silu(round(gate))) * round(up))  (gpt-fast/model.py:258-259)
                float gv[KR], uv[KR], x[KR];
                rounded_row<BF16, KR>(pr.gu, pr.gu_split, (size_t)2 * Z, (size_t)m, gv);
                rounded_row<BF16, KR>(pr.gu, pr.gu_split, (size_t)2 * Z, (size_t)Z + m, uv);
#pragma unroll
                for (int s_ = 0; s_ < KR; ++s_) {
                    const float sl = bits_to_float(float_to_bits<BF16>(gv[s_] / (1.0f + expf(-gv[s_]))), BF16);
                    x[s_] = s_ < pr.T ? bits_to_float(float_to_bits<BF16>(sl * uv[s_]), BF16) : 0.0f;
                }
                store_row<BF16, KR>(dst, x);
            }
        }
        __syncthreads();
        // rows past the phase's last group (GUARD: only the last, partial pair of batches) repeat it; their activations read as zero
        auto issue = [&](u32x2 (&w)[U], const int j0, auto guard) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = decltype(guard)::value ? min(j0 + u, njp - 1) : j0 + u;
                w[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wph + (size_t)j * stride) + (uint32_t)lane);
            }
        };
        auto consume_batch = [&](const u32x2 (&w)[U], const int j0, auto guard) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = decltype(guard)::value ? min(j0 + u, njp - 1) : j0 + u;
                u32x4 xv[KW];
#pragma unroll
                for (int q = 0; q < KW; ++q) {
                    xv[q] = xs[(j * 16 + wave) * KW + q];  // wave-uniform address: one LDS broadcast read per word
                    if (decltype(guard)::value && j0 + u >= njp) xv[q] = (u32x4){0u, 0u, 0u, 0u};
                }
                consume(w[u], xv);
            }
        };
        // software pipeline, two batches of U rows in flight (scheduling barriers: left alone, the machine scheduler sinks every
        // load to just above its first use to save registers — one load in flight per wave and an `s_waitcnt vmcnt(0)` per row)
        constexpr std::false_type full{};
        constexpr std::true_type guarded{};
        u32x2 wa[U], wb[U];
        int j0 = 0;
        if (nfull > 0) {
            issue(wa, 0, full);
            __builtin_amdgcn_sched_barrier(0);
            for (int it = 0; it < nfull; ++it, j0 += 2 * U) {
                issue(wb, j0 + U, full);
                __builtin_amdgcn_sched_barrier(0);
                consume_batch(wa, j0, full);
                __builtin_amdgcn_sched_barrier(0);
                if (it + 1 < nfull) issue(wa, j0 + 2 * U, full);
                __builtin_amdgcn_sched_barrier(0);
                consume_batch(wb, j0 + U, full);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (j0 < njp) {  // the partial pair: one guarded batch if U rows or fewer remain, else two
            issue(wa, j0, guarded);
            const bool two = j0 + U < njp;
            if (two) issue(wb, j0 + U, guarded);
            __builtin_amdgcn_sched_barrier(0);
            consume_batch(wa, j0, guarded);
            if (two) consume_batch(wb, j0 + U, guarded);
        }
    }
    // the 16 waves in fixed order through LDS, one token pair per round
    const uint32_t col_base = (uint32_t)tile * BN;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CPL; ++c) *reinterpret_cast<f32x2*>(red + ((size_t)wave * BN + lane * CPL + c) * 2) = acc[c][p];
        __syncthreads();
        if (tid < BN * 2) {
            const int col = tid >> 1, e = tid & 1;
            float sum = 0.0f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) sum += red[((size_t)w * BN + col) * 2 + e];
            slabs[((size_t)slice * n_total + col_base + col) * KR + 2 * p + e] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// h = [embedding rows of the tokens | ht_in] (+ round(sum slabs));  x = RMSNorm(h) * w  (gpt-fast/model.py:158-161, 289-291).
// Two launches of dim / 256 workgroups, one column per thread (a single workgroup pulls the 2 MB of a 16-slice projection's slabs
// through ONE CU: 30 us, first build):
//   prefill_resid_kernel  h -> ht_out [dim][8], and each workgroup's sums of h^2 per token -> sumsq[workgroup][8]
//   prefill_norm_kernel   every wave adds the workgroups' sums itself (lane = workgroup), x -> xt_out [dim][8] and, optionally,
//                         the normalised vector of token `last` as a plain [dim] vector (the lm_head of the prompt's last token)
// ------------------------------------------------------------------------------------------------
template <bool BF16, int KR>
__global__ __launch_bounds__(256) void prefill_resid_kernel(const uint16_t* __restrict__ emb, const int32_t* __restrict__ tokens, const int T,
                                                            const uint16_t* __restrict__ ht_in, const float* __restrict__ slabs,
                                                            const int split, const int dim, uint16_t* __restrict__ ht_out,
                                                            float* __restrict__ sumsq) {
    __shared__ float part[4][KR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = blockIdx.x * 256 + tid;
    float h[KR];
#pragma unroll
    for (int s = 0; s < KR; ++s) h[s] = 0.0f;
    if (col < dim) {
        if (tokens) {
#pragma unroll
            for (int s = 0; s < KR; ++s) h[s] = bits_to_float(emb[(size_t)tokens[min(s, T - 1)] * dim + col], BF16);
        } else {
            load_row<BF16, KR>(ht_in + (size_t)col * KR, h);
        }
        if (split > 0) {
            float y[KR];
            rounded_row<BF16, KR>(slabs, split, (size_t)dim, (size_t)col, y);
#pragma unroll
            for (int s = 0; s < KR; ++s) h[s] = bits_to_float(float_to_bits<BF16>(h[s] + y[s]), BF16);
        }
#pragma unroll
        for (int s = 0; s < KR; ++s) h[s] = s < T ? h[s] : 0.0f;
        store_row<BF16, KR>(ht_out + (size_t)col * KR, h);
    }
#pragma unroll
    for (int s = 0; s < KR; ++s) {
        const float w = wave_sum_f(h[s] * h[s]);
        if (lane == 0) part[wave][s] = w;
    }
    __syncthreads();
    if (tid < KR) sumsq[(size_t)blockIdx.x * KR + tid] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
}

template <bool BF16, int KR>
__global__ __launch_bounds__(256) void prefill_norm_kernel(const uint16_t* __restrict__ ht, const float* __restrict__ sumsq, const int nwg,
                                                           const uint16_t* __restrict__ norm_w, const float eps, const int dim, const int T,
                                                           uint16_t* __restrict__ xt_out, uint16_t* __restrict__ x_last, const int last) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int col = blockIdx.x * 256 + tid;
    float rstd[KR];
    rstd_rows<KR>(sumsq, nwg, lane, dim, eps, rstd);
    if (col >= dim) return;
    const float nw = bits_to_float(norm_w[col], BF16);
    float x[KR];
    load_row<BF16, KR>(ht + (size_t)col * KR, x);
#pragma unroll
    for (int s = 0; s < KR; ++s) {
        const float xn = bits_to_float(float_to_bits<BF16>(x[s] * rstd[s]), BF16);
        x[s] = s < T ? bits_to_float(float_to_bits<BF16>(xn * nw), BF16) : 0.0f;
    }
    if (xt_out) store_row<BF16, KR>(xt_out + (size_t)col * KR, x);
    if (x_last) {
        float xl = 0.0f;
#pragma unroll
        for (int s = 0; s < KR; ++s) xl = s == last ? x[s] : xl;
        x_last[col] = float_to_bits<BF16>(xl);
    }
}

// ------------------------------------------------------------------------------------------------
// Attention of the prompt's T <= KR tokens at positions 0 .. T-1 (gpt-fast/model.py:170-186): q | k | v from the slabs of the wqkv
// launch, RoPE(q, k), cache rows 0 .. T-1, causal softmax(q K^T / sqrt(d)) V.  One workgroup per query head, thread d = column d
// of the head (all T tokens in registers); the first query head of a KV group writes the cache rows.  yt[n_head * hd][KR].
// ------------------------------------------------------------------------------------------------
template <bool BF16, int HD, int KR>
__global__ __launch_bounds__(HD) void prefill_attention_kernel(const float* __restrict__ slabs, const int split, const int n_head,
                                                               const int n_kv, const int T, const uint16_t* __restrict__ rope,
                                                               uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                                               const int max_seq, const float scale, uint16_t* __restrict__ yt) {
    __shared__ float qs[KR][HD + 1], ks[KR][HD + 1], sc[KR][KR], ls[KR];
    const int h = blockIdx.x, d = threadIdx.x, rep = n_head / n_kv, kvh = h / rep;
    const size_t nq = (size_t)n_head * HD, nkv = (size_t)n_kv * HD, ntot = nq + 2 * nkv;
    float q[KR], k[KR], v[KR];
    {   // the three columns' slabs together: UU slices x three columns of loads in flight per round, slice order kept
        constexpr int NW = KR / 4, UU = KR == 8 ? 4 : 2;
        const size_t cols[3] = {(size_t)h * HD + d, nq + (size_t)kvh * HD + d, nq + nkv + (size_t)kvh * HD + d};
        f32x4 sa[3][NW];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int w = 0; w < NW; ++w) sa[c][w] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < split; k0 += UU) {
            f32x4 va[UU][3][NW];
#pragma unroll
            for (int u = 0; u < UU; ++u)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* p = slabs + ((size_t)min(k0 + u, split - 1) * ntot + cols[c]) * KR;
#pragma unroll
                    for (int w = 0; w < NW; ++w) va[u][c][w] = *reinterpret_cast<const f32x4*>(p + 4 * w);
                }
#pragma unroll
            for (int u = 0; u < UU; ++u)
                if (k0 + u < split) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int w = 0; w < NW; ++w) sa[c][w] += va[u][c][w];
                }
        }
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                q[4 * w + s] = bits_to_float(float_to_bits<BF16>(sa[0][w][s]), BF16);
                k[4 * w + s] = bits_to_float(float_to_bits<BF16>(sa[1][w][s]), BF16);
                v[4 * w + s] = bits_to_float(float_to_bits<BF16>(sa[2][w][s]), BF16);
            }
    }
#pragma unroll
    for (int s = 0; s < KR; ++s) {  // (token slots past T hold whatever the slabs held: keep them out of every sum)
        q[s] = s < T ? q[s] : 0.0f;
        k[s] = s < T ? k[s] : 0.0f;
        v[s] = s < T ? v[s] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < KR; ++s) {
        const uint32_t cs = *reinterpret_cast<const uint32_t*>(rope + ((size_t)min(s, max_seq - 1) * (HD / 2) + (d >> 1)) * 2);
        const float c = bits_to_float(cs & 0xFFFFu, BF16), sn = bits_to_float(cs >> 16, BF16);
        const float qp = __shfl_xor(q[s], 1), kp = __shfl_xor(k[s], 1);
        q[s] = bits_to_float(float_to_bits<BF16>((d & 1) ? rope_odd(qp, q[s], c, sn) : rope_even(q[s], qp, c, sn)), BF16);
        k[s] = bits_to_float(float_to_bits<BF16>((d & 1) ? rope_odd(kp, k[s], c, sn) : rope_even(k[s], kp, c, sn)), BF16);
        qs[s][d] = q[s];
        ks[s][d] = k[s];
        if (s < T && s < max_seq && h % rep == 0) {
            k_cache[((size_t)kvh * max_seq + s) * HD + d] = float_to_bits<BF16>(k[s]);
            v_cache[((size_t)kvh * max_seq + s) * HD + d] = float_to_bits<BF16>(v[s]);
        }
    }
    __syncthreads();
    for (int pq = d; pq < KR * KR; pq += HD) {  // one (query, key) pair per thread and round: the causal half only
        const int s = pq / KR, t = pq % KR;
        float a = 0.0f;
        if (t <= s && s < T) {
            for (int j = 0; j < HD; ++j) a = fmaf(qs[s][j], ks[t][j], a);
            a = bits_to_float(float_to_bits<BF16>(a * scale), BF16);
        }
        sc[s][t] = a;
    }
    __syncthreads();
    if (d < KR) {
        const int s = d;
        float mx = -INFINITY, l = 0.0f;
        for (int t = 0; t <= s; ++t) mx = fmaxf(mx, sc[s][t]);
        for (int t = 0; t < KR; ++t) {
            const float e = t <= s ? expf(sc[s][t] - mx) : 0.0f;
            sc[s][t] = e;
            l += e;
        }
        ls[s] = l;
    }
    __syncthreads();
    float o[KR];
#pragma unroll
    for (int s = 0; s < KR; ++s) {
        float a = 0.0f;
#pragma unroll
        for (int t = 0; t < KR; ++t) a = fmaf(sc[s][t], v[t], a);
        o[s] = s < T ? bits_to_float(float_to_bits<BF16>(a / ls[s]), BF16) : 0.0f;
    }
    store_row<BF16, KR>(yt + ((size_t)h * HD + d) * KR, o);
}

}  // namespace teal

using namespace teal;

extern "C" {

int teal_prefill_gemm(const teal_prefill_in_t* in, const void* w0T, int ld0, int n0, const void* w1T, int ld1, int n1, float* slabs,
                      size_t slabs_bytes, int Z, int T, int dtype, int* split_out, void* stream) {
    if (!in || !w0T || !slabs || !split_out || Z <= 0 || n0 <= 0 || n1 < 0 || (n1 > 0 && !w1T)) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (T < 1 || T > kRowsMax || (Z & 255) || Z > 65536 || (ld0 & 7) || (ld1 & 7) || ld0 < n0 || (n1 > 0 && ld1 < n1)) return TEAL_ERR_SHAPE;
    const int kr = rows_for(T);
    PrefillProd pr = {};
    pr.T = T;
    switch (in->mode) {
        case TEAL_PREFILL_IN_XT:
            if (!in->xt || !aligned16(in->xt)) return TEAL_ERR_ARG;
            pr.xt = reinterpret_cast<const uint16_t*>(in->xt);
            break;
        case TEAL_PREFILL_IN_NORM:
            if (!in->xt || !aligned16(in->xt) || !in->sumsq || !aligned16(in->sumsq) || !in->norm_w || in->nwg < 1 || in->nwg > 64) return TEAL_ERR_ARG;
            pr.xt = reinterpret_cast<const uint16_t*>(in->xt);
            pr.sumsq = in->sumsq; pr.nwg = in->nwg; pr.norm_w = reinterpret_cast<const uint16_t*>(in->norm_w); pr.eps = in->eps;
            break;
        case TEAL_PREFILL_IN_SILU_MUL:
            if (!in->gu_slabs || !aligned16(in->gu_slabs) || in->gu_split < 1 || in->gu_split > kPrefillMaxSplit) return TEAL_ERR_ARG;
            pr.gu = in->gu_slabs; pr.gu_split = in->gu_split;
            if (in->gu_slabs == slabs) return TEAL_ERR_ARG;  // the launch reads its producer's slabs while it writes its own
            break;
        default: return TEAL_ERR_ARG;
    }
    if (!aligned16(w0T) || (w1T && !aligned16(w1T)) || !aligned16(slabs)) return TEAL_ERR_ALIGN;
    DeviceCtx* ctx = device_ctx();
    if (!ctx) return TEAL_ERR_NO_DEVICE;
    const int ncu = ctx->num_cu, ntot = n0 + n1;
    // 256-column tiles; the 16-row groups are dealt to `split` slices: never more workgroups than CUs (a 16-wave workgroup owns its
    // CU: 86 tiles x 3 = 258 workgroups ran 62 us, x 2 = 172 run 38), every wave keeping at least one full pair of batches
    // (split <= Z / 256), and among the candidates the one whose waves stream the fewest 8-row batches (ties: the fewer slabs)
    constexpr int bn = 256;
    if (n0 % bn || n1 % bn) return TEAL_ERR_SHAPE;
    const int tiles = ntot / bn, ngroups = Z >> 4;
    int smax = ncu / tiles;
    if (smax > Z / 256) smax = Z / 256;
    if (smax > kPrefillMaxSplit) smax = kPrefillMaxSplit;
    if (smax < 1) smax = 1;
    int split = 1, best = INT_MAX;
    for (int c = 1; c <= smax; ++c) {
        const int batches = ((ngroups + c - 1) / c + 7) / 8;
        if (batches < best) { best = batches; split = c; }
    }
    if (slabs_bytes < (size_t)split * ntot * kr * sizeof(float)) return TEAL_ERR_WORKSPACE;
    const dim3 grid(tiles, split), block(1024);
    const size_t lds = (size_t)16 * bn * 2 * sizeof(float);  // 32 KB: the activation rows of a phase, then the reduction tile
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int np = (T + 1) / 2, tiles0 = n0 / bn;
    auto* a = reinterpret_cast<const uint16_t*>(w0T);
    auto* b = reinterpret_cast<const uint16_t*>(w1T);
#define TEAL_PG(BF, NPV, PR, KRV) hipLaunchKernelGGL((prefill_gemm_kernel<BF, NPV, PR, KRV>), grid, block, lds, st, pr, a, ld0, b, ld1, tiles0, slabs, Z, ntot)
#define TEAL_PG_PR(BF, NPV, KRV) do { if (in->mode == TEAL_PREFILL_IN_NORM) TEAL_PG(BF, NPV, 1, KRV); else if (in->mode == TEAL_PREFILL_IN_SILU_MUL) TEAL_PG(BF, NPV, 2, KRV); else TEAL_PG(BF, NPV, 0, KRV); } while (0)
#define TEAL_PG_NP(BF) do { switch (np) { case 1: TEAL_PG_PR(BF, 1, 8); break; case 2: TEAL_PG_PR(BF, 2, 8); break; case 3: TEAL_PG_PR(BF, 3, 8); break; \
    case 4: TEAL_PG_PR(BF, 4, 8); break; case 5: TEAL_PG_PR(BF, 5, 16); break; case 6: TEAL_PG_PR(BF, 6, 16); break; case 7: TEAL_PG_PR(BF, 7, 16); break; \
    default: TEAL_PG_PR(BF, 8, 16); } } while (0)
    if (dtype == TEAL_BF16) TEAL_PG_NP(true); else TEAL_PG_NP(false);
#undef TEAL_PG_NP
#undef TEAL_PG_PR
#undef TEAL_PG
    *split_out = split;
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_prefill_resid_norm(const void* emb, const int32_t* tokens, int T, const void* ht_in, const float* slabs, int split,
                            const void* norm_w, float eps, int dim, void* ht_out, void* xt_out, void* x_last, float* sumsq_scratch,
                            int dtype, void* stream) {
    if ((!tokens) == (!ht_in) || (tokens && !emb) || ((xt_out || x_last) && !norm_w) || !ht_out || !sumsq_scratch || dim <= 0 || split < 0 || (split > 0 && !slabs))
        return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (T < 1 || T > kRowsMax || dim > 16384) return TEAL_ERR_SHAPE;
    if (!device_ctx()) return TEAL_ERR_NO_DEVICE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nwg = (dim + 255) / 256;  // <= 64
    auto* e = reinterpret_cast<const uint16_t*>(emb);
    auto* hi = reinterpret_cast<const uint16_t*>(ht_in);
    auto* ho = reinterpret_cast<uint16_t*>(ht_out);
    auto* nw = reinterpret_cast<const uint16_t*>(norm_w);
    auto* xo = reinterpret_cast<uint16_t*>(xt_out);
    auto* xl = reinterpret_cast<uint16_t*>(x_last);
#define TEAL_PR(BF, KRV) do { \
        hipLaunchKernelGGL((prefill_resid_kernel<BF, KRV>), dim3(nwg), dim3(256), 0, st, e, tokens, T, hi, slabs, split, dim, ho, sumsq_scratch); \
        if (xo || xl) hipLaunchKernelGGL((prefill_norm_kernel<BF, KRV>), dim3(nwg), dim3(256), 0, st, ho, sumsq_scratch, nwg, nw, eps, dim, T, xo, xl, T - 1); } while (0)
    if (dtype == TEAL_BF16) { if (rows_for(T) == 8) TEAL_PR(true, 8); else TEAL_PR(true, 16); }
    else { if (rows_for(T) == 8) TEAL_PR(false, 8); else TEAL_PR(false, 16); }
#undef TEAL_PR
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_prefill_attention(const float* qkv_slabs, int split, const void* rope, void* k_cache, void* v_cache, void* yt, int T, int n_head,
                           int n_kv_head, int head_dim, int max_seq, int dtype, void* stream) {
    if (!qkv_slabs || !rope || !k_cache || !v_cache || !yt || split < 1) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if ((head_dim != 64 && head_dim != 128) || n_head <= 0 || n_kv_head <= 0 || n_head % n_kv_head || T < 1 || T > kRowsMax || max_seq < T)
        return TEAL_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
#define TEAL_PA(BF, HDV, KRV) hipLaunchKernelGGL((prefill_attention_kernel<BF, HDV, KRV>), dim3(n_head), dim3(HDV), 0, st, qkv_slabs, split, n_head, n_kv_head, T, \
    reinterpret_cast<const uint16_t*>(rope), reinterpret_cast<uint16_t*>(k_cache), reinterpret_cast<uint16_t*>(v_cache), max_seq, scale, reinterpret_cast<uint16_t*>(yt))
#define TEAL_PA_KR(BF, HDV) do { if (rows_for(T) == 8) TEAL_PA(BF, HDV, 8); else TEAL_PA(BF, HDV, 16); } while (0)
    if (dtype == TEAL_BF16) { if (head_dim == 128) TEAL_PA_KR(true, 128); else TEAL_PA_KR(true, 64); }
    else { if (head_dim == 128) TEAL_PA_KR(false, 128); else TEAL_PA_KR(false, 64); }
#undef TEAL_PA_KR
#undef TEAL_PA
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

}  // extern "C"
