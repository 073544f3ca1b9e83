// teal_gemv_int4.hip — sparse GEMV over int4 group-quantised weights (second half).
//
// The reference ships int4-g32/64/128 weight-only linears for its DENSE gpt-fast path only
// (gpt-fast/quantize.py:58-162 group q-params / quantise / dequantise, :359-443 handler, :483-526 WeightOnlyInt4Linear,
// whose forward is a CUDA-only tinygemm op) and lists quantised TEAL as missing (README.md:110).  Semantics restated:
//     w[n][m] = (q[n][m] - 8) * scale[m / G][n] + zero[m / G][n]        q in 0..15, scale / zero bf16, G = group size
//     y[n]    = sum over kept m of x[m] * w[n][m]                        kept: float32(|x[m]|) > float32(tau), strict
// Layout here (ours to choose — the reference's packed layout is tinygemm's): the column-gathered image of W^T,
// wq[Z][ldb] bytes, byte j of row m = columns 2j (low nibble) and 2j + 1 (high nibble); scales_and_zeros [Z / G][N][2]
// bf16 exactly as the reference stores them (quantize.py:79-93).
//
// Kernel: one 16-wave workgroup = one 128-column tile x one slice of the quantisation groups.  A wave owns whole
// groups; per 32-row unit it ballots the keep mask, deals the kept rows round-robin to its four 16-lane row groups
// (a lane = 8 columns = one dword of a 64-byte row segment), accumulates A = sum x*q and X = sum x per group in fp32
// and applies scale / zero ONCE per (group, column):  y += scale * (A - 8 X) + zero * X  — the group parameters cost
// 32 bytes per lane and group instead of per row.  Split-K over groups is folded into the one launch by arrival
// tickets (the last slice of a tile sums the partials in slice order: deterministic, no atomics on the data).
// HBM-bound skinny GEMV; no MFMA on purpose.
#include "teal_common.h"

#include <limits.h>

namespace teal {


struct Int4Args {
    const uint16_t* x;
    const unsigned char* wq;
    const uint16_t* sz;   // [Z / G][N][2] bf16 (scale, zero)
    uint16_t* y;
    float* ws;            // [ncols][ws_stride] partials (split > 1)
    unsigned* ticket;
    int Z, N, ldb, G;     // ldb: row stride of wq in bytes
    int seg_tile1, seg_tile2;
    float tau0, tau1, tau2;
    int ws_stride;
};

template <bool BF16>
__global__ __launch_bounds__(1024) void sparse_gemv_int4_kernel(const Int4Args a) {
    constexpr int WAVES = 16, BN = 128;
    __shared__ float red[WAVES * BN];
    __shared__ float tflag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, slice = blockIdx.y, split = gridDim.y;
    const int rs = lane >> 4, cl = lane & 15;  // row group of the wave, 4-byte column slot of the tile
    int s = 0;
    if (tile >= a.seg_tile1) s = 1;
    if (tile >= a.seg_tile2) s = 2;
    const float tau = s == 0 ? a.tau0 : (s == 1 ? a.tau1 : a.tau2);
    const int ngroups = a.Z / a.G, upg = a.G / 32;  // 32-row units per group
    const uint32_t col0 = (uint32_t)tile * BN + cl * 8;
    const unsigned char* wp = a.wq + (size_t)tile * (BN / 2) + cl * 4;
    float total[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) total[k] = 0.0f;
    // groups of this workgroup's slice, dealt to the waves round-robin
    for (int gq = slice + split * wave; gq < ngroups; gq += split * WAVES) {
        float A[8], X = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) A[k] = 0.0f;
        // group parameters: requested now, used after the rows (32 B per lane)
        const u32x4* szp = reinterpret_cast<const u32x4*>(a.sz + ((size_t)gq * a.N + col0) * 2);
        const u32x4 sz0 = szp[0], sz1 = szp[1];
        for (int u = 0; u < upg; ++u) {
            const int row0 = gq * a.G + u * 32;
            const float xv = bits_to_float(a.x[row0 + (lane & 31)], BF16);
            uint32_t mask = (uint32_t)__ballot(keep_rule(xv, tau) || (xv != xv));  // lanes 32..63 mirror 0..31
            // up to 8 rounds of 4 kept rows; every load of the unit is issued before the first use
            uint32_t d[8];
            float xr[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int b[4];
                bool ok[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ok[j] = mask != 0u;
                    b[j] = ok[j] ? __builtin_ctz(mask) : 0;
                    mask &= mask - 1u;
                }
                const int br = rs == 0 ? b[0] : (rs == 1 ? b[1] : (rs == 2 ? b[2] : b[3]));
                const bool okr = rs == 0 ? ok[0] : (rs == 1 ? ok[1] : (rs == 2 ? ok[2] : ok[3]));
                const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), b[0]));
                const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), b[1]));
                const float x2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), b[2]));
                const float x3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), b[3]));
                xr[r] = okr ? (rs == 0 ? x0 : (rs == 1 ? x1 : (rs == 2 ? x2 : x3))) : 0.0f;
                d[r] = 0x88888888u;  // (a row that is not there: its x is 0, so it adds nothing to A or X)
                if (okr) d[r] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(wp + (size_t)(row0 + br) * a.ldb));
            }
            // The launch is bound by vector-ALU issue, not by memory (25 us for 17 MB at 7B shapes): nibble -> float as
            // bfe + cvt + fma costs 3 instructions per weight.  Two nibbles at a time become halves without a conversion:
            // (d >> 4j) & 0x000F000F under the exponent bits 0x6400 is the pair {1024 + q_j, 1024 + q_(j+4)} (the byte trick of
            // the int8 kernel, teal_gemv_kernel.h), consumed by the mixed-precision FMA; the constant leaves with X below.
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x2 hq = __builtin_bit_cast(f16x2, ((d[r] >> (4 * j)) & 0x000F000Fu) | 0x64006400u);
                    A[j] = fmaf((float)hq.x, xr[r], A[j]);
                    A[j + 4] = fmaf((float)hq.y, xr[r], A[j + 4]);
                }
                X += xr[r];
            }
        }
        // scale / zero once per (group, column): y += scale * (A - 8 X) + zero * X
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t pr = k < 4 ? sz0[k] : sz1[k - 4];  // bf16 pair: scale (low half), zero (high half)
            const float sc = __uint_as_float(pr << 16), zr = __uint_as_float(pr & 0xFFFF0000u);
            total[k] += sc * (A[k] - 1032.0f * X) + zr * X;  // A carries 1024 + q: (q - 8) = (1024 + q) - 1032
        }
    }
    // reduce: the four row groups of the wave, then the waves in fixed order
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        total[k] += __shfl_xor(total[k], 16);
        total[k] += __shfl_xor(total[k], 32);
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) red[wave * BN + lane * 8 + k] = total[k];
    }
    __syncthreads();
    const uint32_t c = (uint32_t)tile * BN + tid;
    float sum = 0.0f;
    if (tid < BN) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sum += red[w * BN + tid];
        if (split == 1) a.y[c] = float_to_bits<BF16>(sum);
        else __hip_atomic_store(&a.ws[c * (uint32_t)a.ws_stride + slice], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (split > 1) {  // arrival tickets: see gemv_fast_kernel
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
                acc += __hip_atomic_load(&a.ws[c * (uint32_t)a.ws_stride + sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.y[c] = float_to_bits<BF16>(acc);
            if (tid == 0) __hip_atomic_store(&a.ticket[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace teal

using namespace teal;

extern "C" int teal_sparse_qkv_gemv_i4(const void* x, const void* wq, const void* scales_and_zeros, void* y, float tau_q,
                                       float tau_k, float tau_v, int Z, int N, int N_q, int N_kv, int ldb, int groupsize,
                                       int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !wq || !scales_and_zeros || !y || Z <= 0 || N <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (groupsize != 32 && groupsize != 64 && groupsize != 128 && groupsize != 256) return TEAL_ERR_ARG;
    if ((N % 128) || (Z % groupsize) || Z > 65536 || ldb < N / 2 || (ldb & 3)) return TEAL_ERR_SHAPE;
    if (N_q <= 0 || N_kv < 0 || N_q + 2 * N_kv != N || (N_q % 128) || (N_kv % 128)) return TEAL_ERR_SHAPE;
    if (!aligned16(scales_and_zeros) || (reinterpret_cast<uintptr_t>(wq) & 3u)) return TEAL_ERR_ALIGN;
    DeviceCtx* dc = device_ctx();
    if (!dc) return TEAL_ERR_NO_DEVICE;
    const int ntiles = N / 128, ngroups = Z / groupsize;
    int split = dc->num_cu / ntiles;
    if (split > 8) split = 8;
    if (split * 16 > ngroups) split = ngroups / 16;  // every wave of every slice owns at least one group
    if (split < 1) split = 1;
    Int4Args a = {};
    a.x = reinterpret_cast<const uint16_t*>(x);
    a.wq = reinterpret_cast<const unsigned char*>(wq);
    a.sz = reinterpret_cast<const uint16_t*>(scales_and_zeros);
    a.y = reinterpret_cast<uint16_t*>(y);
    a.Z = Z; a.N = N; a.ldb = ldb; a.G = groupsize;
    a.tau0 = tau_q; a.tau1 = tau_k; a.tau2 = tau_v;
    a.seg_tile1 = N_kv > 0 ? N_q / 128 : INT_MAX;
    a.seg_tile2 = N_kv > 0 ? (N_q + N_kv) / 128 : INT_MAX;
    if (split > 1) {
        // split-K over the groups needs the arrival counters of a prepared workspace (teal_workspace_init); without one
        // the launch keeps every group of a tile in one workgroup
        if (!ws_prepared(ws, ws_bytes) || ntiles > kTicketTiles) split = 1;
        else {
            a.ws_stride = (split + 3) & ~3;
            if (ws_bytes - kWsHeaderBytes < (size_t)a.ws_stride * N * sizeof(float)) return TEAL_ERR_WORKSPACE;
            if (!aligned16(ws)) return TEAL_ERR_ALIGN;
            a.ws = ws_slabs(ws);
            a.ticket = ws_tickets(ws);
        }
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(ntiles, split), block(1024);
    if (dtype == TEAL_BF16) hipLaunchKernelGGL((sparse_gemv_int4_kernel<true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((sparse_gemv_int4_kernel<false>), grid, block, 0, st, a);
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}
