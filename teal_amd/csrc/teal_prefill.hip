// teal_prefill.hip — the DENSE prompt pass for short prompts (T <= 8 tokens), hand-fused for gfx950 / CDNA4, wave64.
//
// The reference's prefill is dense by construction: its ops fall back to torch.matmul when the sequence is longer than one
// token (kernels/sparse_gemv.py:271,298), and the rest of the prompt pass is the stock gpt-fast model (gpt-fast/model.py:
// 107-121 forward, 158-161 block, 170-186 attention, 258-259 feed-forward, 289-291 RMSNorm).  The reference's tokens/sec
// counts that pass (gpt-fast/generate.py:458,487-496), and on MI355X an op-by-op prompt pass of a 6-token prompt costs as much
// as nine decode steps (~400 launches of a few microseconds each; profiles/r05_generate_breakdown_before.txt: 20 ms eager,
// 10 ms replayed from a hipGraph).  Here one layer is eight launches over the SAME weight images the decode step streams:
//
//   gemm(wqkv) -> attention (RoPE, cache rows 0..T-1, causal softmax) -> gemm(wo) -> resid_norm -> gemm(w1 | w3) ->
//   silu_mul -> gemm(w2) -> resid_norm
//
// Every hand-over between launches is TRANSPOSED: [feature][8] — the up to eight tokens of a feature are one 16-byte word
// (16-bit activations) or one 32-byte pair (fp32 split-K slabs) — so that a GEMM lane fetches "row m of every token" with one
// load and the GEMM needs no LDS staging and no barrier in its loop.  The GEMM is bound by HBM like the GEMV (every weight
// byte is read once for all tokens: 2 * T flops per byte, far from MFMA territory at T <= 8, and the reduction dimension is
// the strided one in the W^T image, which rules the matrix cores' operand layout out without a second copy of the weights).
// Rounding points are those of the module path's 16-bit tensors (projection outputs, RoPE, attention output, residual adds,
// RMSNorm twice, silu, product); sums are fp32.  Floating-point parity is against the module path, tolerance in the tests.
#include "teal_common.h"

namespace teal {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kRows = 8;  // tokens per transposed word

// ------------------------------------------------------------------------------------------------
// slabs[slice][n][8] (fp32) = sum over the slice's rows m of W^T[m][n] * x[token][m], for up to 2 NP tokens.
//   xt     [Z][8] 16-bit: xt[m][s] = activation m of token s
//   w0/w1  one or two W^T images [Z][ld]: column tiles < tiles0 stream w0, the others w1 (gate | up in one launch)
//   grid (column tiles, row slices); chunk c (64 rows) belongs to slice c mod split, inside the slice to wave (c div split) mod 16
// A lane owns FOUR columns (one 8-byte weight load per row: with eight, the 16 NP accumulators plus two batches of loads in
// flight do not fit the 128 registers a 16-wave workgroup leaves a lane) and 2 NP accumulators per column; LPR lanes cover a row
// segment of BN = 4 LPR columns — 128 or 256 contiguous bytes, what the memory system sees is the GEMV's request — and 64 / LPR
// rows are in flight per wave step.  Accumulation as packed fp32 pairs over the tokens (v_pk_fma_f32).
// ------------------------------------------------------------------------------------------------
template <bool BF16, int LPR, int NP>
__global__ __launch_bounds__(1024) void prefill_gemm_kernel(const uint16_t* __restrict__ xt, const uint16_t* __restrict__ w0, const int ld0,
                                                            const uint16_t* __restrict__ w1, const int ld1, const int tiles0,
                                                            float* __restrict__ slabs, const int Z, const int n_total) {
    constexpr int WAVES = 16, CPL = 4, RPW = 64 / LPR, BN = LPR * CPL, STEPS = 64 / RPW, U = 4;
    static_assert(STEPS % (2 * U) == 0, "whole pairs of batches per chunk");
    extern __shared__ __align__(16) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [WAVES][BN][4]: two token pairs per reduction round
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, slice = blockIdx.y, split = gridDim.y;
    const int g = lane / LPR, cl = lane % LPR;
    const bool second = tile >= tiles0;
    const uint16_t* wp = (second ? w1 : w0) + (size_t)(second ? tile - tiles0 : tile) * BN + cl * CPL;
    const uint32_t ld = (uint32_t)(second ? ld1 : ld0);
    const int nch = Z >> 6;
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
    auto issue = [&](u32x2 (&w)[U], u32x4 (&x)[U], const uint32_t m0, const int batch) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t m = m0 + (uint32_t)(batch * U + u) * RPW;
            w[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wp + (size_t)m * ld));
            x[u] = *reinterpret_cast<const u32x4*>(xt + (size_t)m * kRows);
        }
    };
    // the wave's chunks: c = slice + split * (wave + 16 k); two batches of U rows per lane in flight
    for (int c = slice + split * wave; c < nch; c += split * WAVES) {
        const uint32_t m0 = (uint32_t)c * 64u + g;
        u32x2 wa[U], wb[U];
        u32x4 xa[U], xb[U];
        issue(wa, xa, m0, 0);
#pragma unroll
        for (int b = 0; b < STEPS / U; b += 2) {
            issue(wb, xb, m0, b + 1);
#pragma unroll
            for (int u = 0; u < U; ++u) consume(wa[u], xa[u]);
            if (b + 2 < STEPS / U) issue(wa, xa, m0, b + 2);
#pragma unroll
            for (int u = 0; u < U; ++u) consume(wb[u], xb[u]);
        }
    }
    // row groups of the wave (butterflies), then the 16 waves in fixed order through LDS, two token pairs per round
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float v = acc[c][p][e];
                if constexpr (LPR <= 16) v = xor_add<16>(v);
                v = xor_add<32>(v);
                acc[c][p][e] = v;
            }
    const uint32_t col_base = (uint32_t)tile * BN;
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += 2) {
        if (p0) __syncthreads();
        if (lane < LPR) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                f32x4 v = {acc[c][p0][0], acc[c][p0][1], 0.0f, 0.0f};
                if (p0 + 1 < NP) { v[2] = acc[c][p0 + 1][0]; v[3] = acc[c][p0 + 1][1]; }
                *reinterpret_cast<f32x4*>(red + ((size_t)wave * BN + lane * CPL + c) * 4) = v;
            }
        }
        __syncthreads();
        if (tid < BN * 4) {
            const int col = tid >> 2, sv = tid & 3;
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s += red[((size_t)w * BN + col) * 4 + sv];
            if (2 * p0 + sv < 2 * NP) slabs[((size_t)slice * n_total + col_base + col) * kRows + 2 * p0 + sv] = s;
        }
    }
}

// sum of `split` slabs in slice order, rounded once to the activation dtype: what a projection's 16-bit output tensor holds
template <bool BF16>
__device__ __forceinline__ void rounded_row(const float* __restrict__ slabs, const int split, const size_t n_total, const size_t col,
                                            float (&out)[kRows]) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < split; ++k) {
        const float* p = slabs + ((size_t)k * n_total + col) * kRows;
        a += *reinterpret_cast<const f32x4*>(p);
        b += *reinterpret_cast<const f32x4*>(p + 4);
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
// h = [embedding rows of the tokens | ht_in] (+ round(sum slabs));  x = RMSNorm(h) * w  (gpt-fast/model.py:158-161, 289-291).
// One workgroup: thread t owns columns t, t + 1024, ... for all eight tokens.  Writes ht_out [dim][8], xt_out [dim][8] and,
// optionally, the normalised vector of token `last` as a plain [dim] vector (the lm_head of the prompt's last token).
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(1024) void prefill_resid_norm_kernel(const uint16_t* __restrict__ emb, const int32_t* __restrict__ tokens,
                                                                  const int T, const uint16_t* __restrict__ ht_in,
                                                                  const float* __restrict__ slabs, const int split,
                                                                  const uint16_t* __restrict__ norm_w, const float eps, const int dim,
                                                                  uint16_t* __restrict__ ht_out, uint16_t* __restrict__ xt_out,
                                                                  uint16_t* __restrict__ x_last, const int last) {
    __shared__ float part[16][kRows];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float ss[kRows];
#pragma unroll
    for (int s = 0; s < kRows; ++s) ss[s] = 0.0f;
    // pass 1: h, stored (the second pass reads this thread's own words back), and the sums of squares
    for (int col = tid; col < dim; col += 1024) {
        float h[kRows];
        if (tokens) {
#pragma unroll
            for (int s = 0; s < kRows; ++s) h[s] = s < T ? bits_to_float(emb[(size_t)tokens[s] * dim + col], BF16) : 0.0f;
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
        for (int s = 0; s < kRows; ++s) { if (s >= T) h[s] = 0.0f; ss[s] = fmaf(h[s], h[s], ss[s]); }
        *reinterpret_cast<u32x4*>(ht_out + (size_t)col * kRows) = pack_row<BF16>(h);
    }
#pragma unroll
    for (int s = 0; s < kRows; ++s) {
        const float w = wave_sum_f(ss[s]);
        if (lane == 0) part[wave][s] = w;
    }
    __syncthreads();
    float rstd[kRows];
#pragma unroll
    for (int s = 0; s < kRows; ++s) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w][s];
        rstd[s] = rsqrtf(t / (float)dim + eps);
    }
    for (int col = tid; col < dim; col += 1024) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(ht_out + (size_t)col * kRows);
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
}

// xt[col][s] = round(round(silu(round(gate))) * round(up))  (gpt-fast/model.py:258-259) from the slabs of the gate | up launch
template <bool BF16>
__global__ __launch_bounds__(256) void prefill_silu_mul_kernel(const float* __restrict__ slabs, const int split, const int inter,
                                                               const int T, uint16_t* __restrict__ xt) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= inter) return;
    float gv[kRows], uv[kRows], x[kRows];
    rounded_row<BF16>(slabs, split, (size_t)2 * inter, (size_t)col, gv);
    rounded_row<BF16>(slabs, split, (size_t)2 * inter, (size_t)inter + col, uv);
#pragma unroll
    for (int s = 0; s < kRows; ++s) {
        const float sl = bits_to_float(float_to_bits<BF16>(gv[s] / (1.0f + expf(-gv[s]))), BF16);
        x[s] = s < T ? bits_to_float(float_to_bits<BF16>(sl * uv[s]), BF16) : 0.0f;
    }
    *reinterpret_cast<u32x4*>(xt + (size_t)col * kRows) = pack_row<BF16>(x);
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
    rounded_row<BF16>(slabs, split, ntot, (size_t)h * HD + d, q);
    rounded_row<BF16>(slabs, split, ntot, nq + (size_t)kvh * HD + d, k);
    rounded_row<BF16>(slabs, split, ntot, nq + nkv + (size_t)kvh * HD + d, v);
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

int teal_prefill_gemm(const void* xt, const void* w0T, int ld0, int n0, const void* w1T, int ld1, int n1, float* slabs,
                      size_t slabs_bytes, int Z, int T, int dtype, int* split_out, void* stream) {
    if (!xt || !w0T || !slabs || !split_out || Z <= 0 || n0 <= 0 || n1 < 0 || (n1 > 0 && !w1T)) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (T < 1 || T > kRows || (Z & 63) || Z > 65536 || (ld0 & 7) || (ld1 & 7) || ld0 < n0 || (n1 > 0 && ld1 < n1)) return TEAL_ERR_SHAPE;
    if (!aligned16(xt) || !aligned16(w0T) || (w1T && !aligned16(w1T)) || !aligned16(slabs)) return TEAL_ERR_ALIGN;
    DeviceCtx* ctx = device_ctx();
    if (!ctx) return TEAL_ERR_NO_DEVICE;
    const int ncu = ctx->num_cu, ntot = n0 + n1;
    // 128-column tiles (256-byte row segments) when they still cover two thirds of the CUs, else 64-column tiles; the rows are
    // sliced (fp32 slabs, summed by the consumer in slice order) until tiles x slices ~ the CU count
    int lpr = 32;
    if (n0 % 128 || n1 % 128 || (ntot / 128) * 3 < ncu * 2) lpr = 16;
    const int bn = lpr * 4;
    if (n0 % bn || n1 % bn) return TEAL_ERR_SHAPE;
    const int tiles = ntot / bn, nch = Z >> 6;
    int split = ncu / tiles;
    if (split > nch / 16) split = nch / 16;
    if (split > 8) split = 8;
    if (split < 1) split = 1;
    if (slabs_bytes < (size_t)split * ntot * kRows * sizeof(float)) return TEAL_ERR_WORKSPACE;
    const dim3 grid(tiles, split), block(1024);
    const size_t lds = (size_t)16 * bn * 4 * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int np = (T + 1) / 2, tiles0 = n0 / bn;
    auto* x = reinterpret_cast<const uint16_t*>(xt);
    auto* a = reinterpret_cast<const uint16_t*>(w0T);
    auto* b = reinterpret_cast<const uint16_t*>(w1T);
#define TEAL_PG(BF, LP, NPV) hipLaunchKernelGGL((prefill_gemm_kernel<BF, LP, NPV>), grid, block, lds, st, x, a, ld0, b, ld1, tiles0, slabs, Z, ntot)
#define TEAL_PG_NP(BF, LP) do { switch (np) { case 1: TEAL_PG(BF, LP, 1); break; case 2: TEAL_PG(BF, LP, 2); break; case 3: TEAL_PG(BF, LP, 3); break; default: TEAL_PG(BF, LP, 4); } } while (0)
#define TEAL_PG_L(BF) do { if (lpr == 32) TEAL_PG_NP(BF, 32); else TEAL_PG_NP(BF, 16); } while (0)
    if (dtype == TEAL_BF16) TEAL_PG_L(true); else TEAL_PG_L(false);
#undef TEAL_PG_L
#undef TEAL_PG_NP
#undef TEAL_PG
    *split_out = split;
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_prefill_resid_norm(const void* emb, const int32_t* tokens, int T, const void* ht_in, const float* slabs, int split,
                            const void* norm_w, float eps, int dim, void* ht_out, void* xt_out, void* x_last, int dtype, void* stream) {
    if ((!tokens) == (!ht_in) || (tokens && !emb) || !norm_w || !ht_out || dim <= 0 || split < 0 || (split > 0 && !slabs)) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (T < 1 || T > kRows || dim > 16384) return TEAL_ERR_SHAPE;
    if (!device_ctx()) return TEAL_ERR_NO_DEVICE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define TEAL_PRN(BF) hipLaunchKernelGGL((prefill_resid_norm_kernel<BF>), dim3(1), dim3(1024), 0, st, reinterpret_cast<const uint16_t*>(emb), tokens, T, \
    reinterpret_cast<const uint16_t*>(ht_in), slabs, split, reinterpret_cast<const uint16_t*>(norm_w), eps, dim, reinterpret_cast<uint16_t*>(ht_out),        \
    reinterpret_cast<uint16_t*>(xt_out), reinterpret_cast<uint16_t*>(x_last), T - 1)
    if (dtype == TEAL_BF16) TEAL_PRN(true); else TEAL_PRN(false);
#undef TEAL_PRN
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

int teal_prefill_silu_mul(const float* gu_slabs, int split, int inter, int T, void* xt, int dtype, void* stream) {
    if (!gu_slabs || !xt || split < 1 || inter <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (T < 1 || T > kRows) return TEAL_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((inter + 255) / 256), block(256);
    if (dtype == TEAL_BF16) hipLaunchKernelGGL((prefill_silu_mul_kernel<true>), grid, block, 0, st, gu_slabs, split, inter, T, reinterpret_cast<uint16_t*>(xt));
    else hipLaunchKernelGGL((prefill_silu_mul_kernel<false>), grid, block, 0, st, gu_slabs, split, inter, T, reinterpret_cast<uint16_t*>(xt));
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
