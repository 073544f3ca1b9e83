// teal_prefill.hip — the DENSE prompt pass for short prompts (T <= 8 tokens), hand-fused for gfx950 / CDNA4, wave64.
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
// Every hand-over between launches is TRANSPOSED: [feature][8] — the up to eight tokens of a feature are one 16-byte word
// (16-bit activations) or one 32-byte pair (fp32 split-K slabs) — so that "row m of every token" is one word: a GEMM workgroup
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
constexpr int kRows = 8;  // tokens per transposed word
constexpr int kPrefillMaxSplit = 16;

// sum of `split` slabs in slice order, rounded once to the activation dtype: what a projection's 16-bit output tensor holds
template <bool BF16>
__device__ __forceinline__ void rounded_row(const float* __restrict__ slabs, const int split, const size_t n_total, const size_t col,
                                            float (&out)[kRows]) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < split; k0 += 4) {  // four slices' loads in flight (clamped: a repeated slice is not added); slice order kept
        f32x4 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* p = slabs + ((size_t)min(k0 + u, split - 1) * n_total + col) * kRows;
            va[u] = *reinterpret_cast<const f32x4*>(p);
            vb[u] = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + u < split) { a += va[u]; b += vb[u]; }
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        out[s] = bits_to_float(float_to_bits<BF16>(a[s]), BF16);
        out[4 + s] = bits_to_float(float_to_bits<BF16>(b[s]), BF16);
    }
}

template <bool BF16>
__device__ __forceinline__ u32x4 pack_row(const float (&v)[kRows]) {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (uint32_t)float_to_bits<BF16>(v[2 * j]) | ((uint32_t)float_to_bits<BF16>(v[2 * j + 1]) << 16);
    return r;
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
// per round.  Accumulation as packed fp32 pairs over the tokens (v_pk_fma_f32).  First build (row groups inside a wave, the
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

template <bool BF16, int NP, int PROD>
__global__ __launch_bounds__(1024) void prefill_gemm_kernel(const PrefillProd pr, const uint16_t* __restrict__ w0, const int ld0,
                                                            const uint16_t* __restrict__ w1, const int ld1, const int tiles0,
                                                            float* __restrict__ slabs, const int Z, const int n_total) {
    const uint16_t* __restrict__ xt = pr.xt;
    constexpr int WAVES = 16, CPL = 4, BN = 256, U = 8, PHASE_GROUPS = 128;  // 128 groups x 16 rows x 16 bytes = 32 KB of activations
    extern __shared__ __align__(16) unsigned char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem);   // the slice's activation rows of the current phase: xs[j * 16 + wave]
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
    auto consume = [&](const u32x2 w, const u32x4 xv) {
        f32x2 xp[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) xp[p] = (f32x2){bits_to_float(xv[p] & 0xFFFFu, BF16), bits_to_float(xv[p] >> 16, BF16)};
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
        [[maybe_unused]] float rstd[kRows];
        if constexpr (PROD == 1) {  // lane = producing workgroup of the resid launch (nwg <= 64); the total in workgroup order
            f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
            if (lane < pr.nwg) {
                pa = *reinterpret_cast<const f32x4*>(pr.sumsq + (size_t)lane * kRows);
                pb = *reinterpret_cast<const f32x4*>(pr.sumsq + (size_t)lane * kRows + 4);
            }
#pragma unroll
            for (int s_ = 0; s_ < kRows; ++s_) rstd[s_] = rsqrtf(wave_sum_f(s_ < 4 ? pa[s_ & 3] : pb[s_ & 3]) / (float)Z + pr.eps);
        }
        for (int r = tid; r < njp * 16; r += 1024) {
            const uint32_t m = (uint32_t)(slice + split * (jb + (r >> 4))) * 16u + (uint32_t)(r & 15);
            if constexpr (PROD == 0) {
                xs[r] = *reinterpret_cast<const u32x4*>(xt + (size_t)m * kRows);
            } else if constexpr (PROD == 1) {  // x = round(round(h * rstd) * w)  (gpt-fast/model.py:289-291)
                const u32x4 v = *reinterpret_cast<const u32x4*>(xt + (size_t)m * kRows);
                const float nw = bits_to_float(pr.norm_w[m], BF16);
                float x[kRows];
#pragma unroll
                for (int j = 0; j < 4; ++j) { x[2 * j] = bits_to_float(v[j] & 0xFFFFu, BF16); x[2 * j + 1] = bits_to_float(v[j] >> 16, BF16); }
#pragma unroll
                for (int s_ = 0; s_ < kRows; ++s_) {
                    const float xn = bits_to_float(float_to_bits<BF16>(x[s_] * rstd[s_]), BF16);
                    x[s_] = s_ < pr.T ? bits_to_float(float_to_bits<BF16>(xn * nw), BF16) : 0.0f;
                }
                xs[r] = pack_row<BF16>(x);
            } else {  // x = round(round(silu(round(gate))) * round(up))  (gpt-fast/model.py:258-259)
                float gv[kRows], uv[kRows], x[kRows];
                rounded_row<BF16>(pr.gu, pr.gu_split, (size_t)2 * Z, (size_t)m, gv);
                rounded_row<BF16>(pr.gu, pr.gu_split, (size_t)2 * Z, (size_t)Z + m, uv);
#pragma unroll
                for (int s_ = 0; s_ < kRows; ++s_) {
                    const float sl = bits_to_float(float_to_bits<BF16>(gv[s_] / (1.0f + expf(-gv[s_]))), BF16);
                    x[s_] = s_ < pr.T ? bits_to_float(float_to_bits<BF16>(sl * uv[s_]), BF16) : 0.0f;
                }
                xs[r] = pack_row<BF16>(x);
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
                u32x4 xv = xs[j * 16 + wave];  // wave-uniform address: one LDS broadcast read
                if (decltype(guard)::value && j0 + u >= njp) xv = (u32x4){0u, 0u, 0u, 0u};
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
        if (j0 < njp) {  // the partial pair: one guarded batch if eight rows or fewer remain, else two
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
            slabs[((size_t)slice * n_total + col_base + col) * kRows + 2 * p + e] = sum;
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
template <bool BF16>
__global__ __launch_bounds__(256) void prefill_resid_kernel(const uint16_t* __restrict__ emb, const int32_t* __restrict__ tokens, const int T,
                                                            const uint16_t* __restrict__ ht_in, const float* __restrict__ slabs,
                                                            const int split, const int dim, uint16_t* __restrict__ ht_out,
                                                            float* __restrict__ sumsq) {
    __shared__ float part[4][kRows];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = blockIdx.x * 256 + tid;
    float h[kRows];
#pragma unroll
    for (int s = 0; s < kRows; ++s) h[s] = 0.0f;
    if (col < dim) {
        if (tokens) {
#pragma unroll
            for (int s = 0; s < kRows; ++s) h[s] = bits_to_float(emb[(size_t)tokens[min(s, T - 1)] * dim + col], BF16);
        } else {
            const u32x4 v = *reinterpret_cast<const u32x4*>(ht_in + (size_t)col * kRows);
#pragma unroll
            for (int j = 0; j < 4; ++j) { h[2 * j] = bits_to_float(v[j] & 0xFFFFu, BF16); h[2 * j + 1] = bits_to_float(v[j] >> 16, BF16); }
        }
        if (split > 0) {
            float y[kRows];
            rounded_row<BF16>(slabs, split, (size_t)dim, (size_t)col, y);
#pragma unroll
            for (int s = 0; s < kRows; ++s) h[s] = bits_to_float(float_to_bits<BF16>(h[s] + y[s]), BF16);
        }
#pragma unroll
        for (int s = 0; s < kRows; ++s) h[s] = s < T ? h[s] : 0.0f;
        *reinterpret_cast<u32x4*>(ht_out + (size_t)col * kRows) = pack_row<BF16>(h);
    }
#pragma unroll
    for (int s = 0; s < kRows; ++s) {
        const float w = wave_sum_f(h[s] * h[s]);
        if (lane == 0) part[wave][s] = w;
    }
    __syncthreads();
    if (tid < kRows) sumsq[(size_t)blockIdx.x * kRows + tid] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
}

template <bool BF16>
__global__ __launch_bounds__(256) void prefill_norm_kernel(const uint16_t* __restrict__ ht, const float* __restrict__ sumsq, const int nwg,
                                                           const uint16_t* __restrict__ norm_w, const float eps, const int dim, const int T,
                                                           uint16_t* __restrict__ xt_out, uint16_t* __restrict__ x_last, const int last) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int col = blockIdx.x * 256 + tid;
    // lane = producing workgroup (nwg = dim / 256 <= 64): its eight sums are two 16-byte loads; the total in workgroup order
    f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
    if (lane < nwg) {
        pa = *reinterpret_cast<const f32x4*>(sumsq + (size_t)lane * kRows);
        pb = *reinterpret_cast<const f32x4*>(sumsq + (size_t)lane * kRows + 4);
    }
    float rstd[kRows];
#pragma unroll
    for (int s = 0; s < kRows; ++s) rstd[s] = rsqrtf(wave_sum_f(s < 4 ? pa[s & 3] : pb[s & 3]) / (float)dim + eps);
    if (col >= dim) return;
    const u32x4 v = *reinterpret_cast<const u32x4*>(ht + (size_t)col * kRows);
    const float nw = bits_to_float(norm_w[col], BF16);
    float x[kRows];
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[2 * j] = bits_to_float(v[j] & 0xFFFFu, BF16); x[2 * j + 1] = bits_to_float(v[j] >> 16, BF16); }
#pragma unroll
    for (int s = 0; s < kRows; ++s) {
        const float xn = bits_to_float(float_to_bits<BF16>(x[s] * rstd[s]), BF16);
        x[s] = s < T ? bits_to_float(float_to_bits<BF16>(xn * nw), BF16) : 0.0f;
    }
    if (xt_out) *reinterpret_cast<u32x4*>(xt_out + (size_t)col * kRows) = pack_row<BF16>(x);
    if (x_last) {
        float xl = 0.0f;
#pragma unroll
        for (int s = 0; s < kRows; ++s) xl = s == last ? x[s] : xl;
        x_last[col] = float_to_bits<BF16>(xl);
    }
}

// ------------------------------------------------------------------------------------------------
// Attention of the prompt's T <= 8 tokens at positions 0 .. T-1 (gpt-fast/model.py:170-186): q | k | v from the slabs of the wqkv
// launch, RoPE(q, k), cache rows 0 .. T-1, causal softmax(q K^T / sqrt(d)) V.  One workgroup per query head, thread d = column d
// of the head (all T tokens in registers); the first query head of a KV group writes the cache rows.  yt[n_head * hd][8].
// ------------------------------------------------------------------------------------------------
template <bool BF16, int HD>
__global__ __launch_bounds__(HD) void prefill_attention_kernel(const float* __restrict__ slabs, const int split, const int n_head,
                                                               const int n_kv, const int T, const uint16_t* __restrict__ rope,
                                                               uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                                               const int max_seq, const float scale, uint16_t* __restrict__ yt) {
    __shared__ float qs[kRows][HD + 1], ks[kRows][HD + 1], sc[kRows][kRows], ls[kRows];
    const int h = blockIdx.x, d = threadIdx.x, rep = n_head / n_kv, kvh = h / rep;
    const size_t nq = (size_t)n_head * HD, nkv = (size_t)n_kv * HD, ntot = nq + 2 * nkv;
    float q[kRows], k[kRows], v[kRows];
    {   // the three columns' slabs together: four slices x three columns of loads in flight per round, slice order kept
        const size_t cols[3] = {(size_t)h * HD + d, nq + (size_t)kvh * HD + d, nq + nkv + (size_t)kvh * HD + d};
        f32x4 sa[3], sb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { sa[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; sb[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        for (int k0 = 0; k0 < split; k0 += 4) {
            f32x4 va[4][3], vb[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* p = slabs + ((size_t)min(k0 + u, split - 1) * ntot + cols[c]) * kRows;
                    va[u][c] = *reinterpret_cast<const f32x4*>(p);
                    vb[u][c] = *reinterpret_cast<const f32x4*>(p + 4);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < split) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { sa[c] += va[u][c]; sb[c] += vb[u][c]; }
                }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            q[s] = bits_to_float(float_to_bits<BF16>(sa[0][s]), BF16); q[4 + s] = bits_to_float(float_to_bits<BF16>(sb[0][s]), BF16);
            k[s] = bits_to_float(float_to_bits<BF16>(sa[1][s]), BF16); k[4 + s] = bits_to_float(float_to_bits<BF16>(sb[1][s]), BF16);
            v[s] = bits_to_float(float_to_bits<BF16>(sa[2][s]), BF16); v[4 + s] = bits_to_float(float_to_bits<BF16>(sb[2][s]), BF16);
        }
    }
#pragma unroll
    for (int s = 0; s < kRows; ++s) {  // (token slots past T hold whatever the slabs held: keep them out of every sum)
        q[s] = s < T ? q[s] : 0.0f;
        k[s] = s < T ? k[s] : 0.0f;
        v[s] = s < T ? v[s] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < kRows; ++s) {
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
    if (d < kRows * kRows) {  // one (query, key) pair per thread: the causal half only
        const int s = d / kRows, t = d % kRows;
        float a = 0.0f;
        if (t <= s && s < T) {
            for (int j = 0; j < HD; ++j) a = fmaf(qs[s][j], ks[t][j], a);
            a = bits_to_float(float_to_bits<BF16>(a * scale), BF16);
        }
        sc[s][t] = a;
    }
    __syncthreads();
    if (d < kRows) {
        const int s = d;
        float mx = -INFINITY, l = 0.0f;
        for (int t = 0; t <= s; ++t) mx = fmaxf(mx, sc[s][t]);
        for (int t = 0; t < kRows; ++t) {
            const float e = t <= s ? expf(sc[s][t] - mx) : 0.0f;
            sc[s][t] = e;
            l += e;
        }
        ls[s] = l;
    }
    __syncthreads();
    float o[kRows];
#pragma unroll
    for (int s = 0; s < kRows; ++s) {
        float a = 0.0f;
#pragma unroll
        for (int t = 0; t < kRows; ++t) a = fmaf(sc[s][t], v[t], a);
        o[s] = s < T ? bits_to_float(float_to_bits<BF16>(a / ls[s]), BF16) : 0.0f;
    }
    *reinterpret_cast<u32x4*>(yt + ((size_t)h * HD + d) * kRows) = pack_row<BF16>(o);
}

}  // namespace teal

using namespace teal;

extern "C" {

int teal_prefill_gemm(const teal_prefill_in_t* in, const void* w0T, int ld0, int n0, const void* w1T, int ld1, int n1, float* slabs,
                      size_t slabs_bytes, int Z, int T, int dtype, int* split_out, void* stream) {
    if (!in || !w0T || !slabs || !split_out || Z <= 0 || n0 <= 0 || n1 < 0 || (n1 > 0 && !w1T)) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (T < 1 || T > kRows || (Z & 255) || Z > 65536 || (ld0 & 7) || (ld1 & 7) || ld0 < n0 || (n1 > 0 && ld1 < n1)) return TEAL_ERR_SHAPE;
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
    if (slabs_bytes < (size_t)split * ntot * kRows * sizeof(float)) return TEAL_ERR_WORKSPACE;
    const dim3 grid(tiles, split), block(1024);
    const size_t lds = (size_t)16 * bn * 2 * sizeof(float);  // 32 KB: the activation rows of a phase, then the reduction tile
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int np = (T + 1) / 2, tiles0 = n0 / bn;
    auto* a = reinterpret_cast<const uint16_t*>(w0T);
    auto* b = reinterpret_cast<const uint16_t*>(w1T);
#define TEAL_PG(BF, NPV, PR) hipLaunchKernelGGL((prefill_gemm_kernel<BF, NPV, PR>), grid, block, lds, st, pr, a, ld0, b, ld1, tiles0, slabs, Z, ntot)
#define TEAL_PG_PR(BF, NPV) do { if (in->mode == TEAL_PREFILL_IN_NORM) TEAL_PG(BF, NPV, 1); else if (in->mode == TEAL_PREFILL_IN_SILU_MUL) TEAL_PG(BF, NPV, 2); else TEAL_PG(BF, NPV, 0); } while (0)
#define TEAL_PG_NP(BF) do { switch (np) { case 1: TEAL_PG_PR(BF, 1); break; case 2: TEAL_PG_PR(BF, 2); break; case 3: TEAL_PG_PR(BF, 3); break; default: TEAL_PG_PR(BF, 4); } } while (0)
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
    if (T < 1 || T > kRows || dim > 16384) return TEAL_ERR_SHAPE;
    if (!device_ctx()) return TEAL_ERR_NO_DEVICE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nwg = (dim + 255) / 256;  // <= 64
    auto* e = reinterpret_cast<const uint16_t*>(emb);
    auto* hi = reinterpret_cast<const uint16_t*>(ht_in);
    auto* ho = reinterpret_cast<uint16_t*>(ht_out);
    auto* nw = reinterpret_cast<const uint16_t*>(norm_w);
    auto* xo = reinterpret_cast<uint16_t*>(xt_out);
    auto* xl = reinterpret_cast<uint16_t*>(x_last);
    if (dtype == TEAL_BF16) {
        hipLaunchKernelGGL((prefill_resid_kernel<true>), dim3(nwg), dim3(256), 0, st, e, tokens, T, hi, slabs, split, dim, ho, sumsq_scratch);
        if (xo || xl) hipLaunchKernelGGL((prefill_norm_kernel<true>), dim3(nwg), dim3(256), 0, st, ho, sumsq_scratch, nwg, nw, eps, dim, T, xo, xl, T - 1);
    } else {
        hipLaunchKernelGGL((prefill_resid_kernel<false>), dim3(nwg), dim3(256), 0, st, e, tokens, T, hi, slabs, split, dim, ho, sumsq_scratch);
        if (xo || xl) hipLaunchKernelGGL((prefill_norm_kernel<false>), dim3(nwg), dim3(256), 0, st, ho, sumsq_scratch, nwg, nw, eps, dim, T, xo, xl, T - 1);
    }
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_prefill_attention(const float* qkv_slabs, int split, const void* rope, void* k_cache, void* v_cache, void* yt, int T, int n_head,
                           int n_kv_head, int head_dim, int max_seq, int dtype, void* stream) {
    if (!qkv_slabs || !rope || !k_cache || !v_cache || !yt || split < 1) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if ((head_dim != 64 && head_dim != 128) || n_head <= 0 || n_kv_head <= 0 || n_head % n_kv_head || T < 1 || T > kRows || max_seq < T)
        return TEAL_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
#define TEAL_PA(BF, HDV) hipLaunchKernelGGL((prefill_attention_kernel<BF, HDV>), dim3(n_head), dim3(HDV), 0, st, qkv_slabs, split, n_head, n_kv_head, T, \
    reinterpret_cast<const uint16_t*>(rope), reinterpret_cast<uint16_t*>(k_cache), reinterpret_cast<uint16_t*>(v_cache), max_seq, scale, reinterpret_cast<uint16_t*>(yt))
    if (dtype == TEAL_BF16) { if (head_dim == 128) TEAL_PA(true, 128); else TEAL_PA(true, 64); }
    else { if (head_dim == 128) TEAL_PA(false, 128); else TEAL_PA(false, 64); }
#undef TEAL_PA
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

}  // extern "C"
