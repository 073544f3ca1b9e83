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
// Kernel: one 16-wave workgroup = one 128-column tile x one slice of the 32-row units.  A wave owns units; per unit it
// ballots the keep mask, compacts the kept rows into its LDS list, deals them to its four 16-lane row groups (a lane = 8
// columns = one dword of a 64-byte row segment), accumulates A = sum x * (1024 + q) and X = sum x in fp32 and applies
// scale / zero ONCE per (unit, column):  y += scale * (A - 1032 X) + zero * X  — the group parameters cost 32 bytes per
// lane and unit instead of per row.  Split-K over units is folded into the one launch by arrival tickets (the last slice
// of a tile sums the partials in slice order: deterministic, no atomics on the data).  No MFMA on purpose.
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

// Round 3 form.  Round 2's kernel walked a wave's groups one after the other — activation load -> ballot -> a serial
// scalar chain (count-trailing-zeros per kept row) dealing the rows to the lane groups -> row loads -> arithmetic — i.e.
// two dependent memory round trips and ~100 scalar instructions per 32-row unit: 24 us per 7B launch for 17 MB.  Now, like
// the 16-bit kernel (teal_gemv_fast.h): a wave handles its 32-row units in PASSES of four; all activations and group
// parameters of the pass leave first; the ballots compact (row, x) pairs into the wave's LDS list with mbcnt ranks (no
// scalar chain); then EVERY row load of the pass is issued before the first is consumed.  scale / zero are applied per
// UNIT with the parameters of the unit's group — y += scale * (A_u - 1032 X_u) + zero * X_u is linear in the units of a
// group — so units are independent whatever the group size (A carries 1024 + q: two nibbles become two halves by one and_or
// under the exponent bits, the byte trick of the int8 kernel).
template <bool BF16>
__global__ __launch_bounds__(1024) void sparse_gemv_int4_kernel(const Int4Args a) {
    constexpr int WAVES = 16, BN = 128, UP = 4;  // UP: units per pass
    __shared__ float red[WAVES * BN];
    __shared__ uint32_t lists[WAVES][UP * 32];
    __shared__ float tflag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, slice = blockIdx.y, split = gridDim.y;
    const int rs = lane >> 4, cl = lane & 15;  // row group of the wave, 4-byte column slot of the tile
    int s = 0;
    if (tile >= a.seg_tile1) s = 1;
    if (tile >= a.seg_tile2) s = 2;
    const float tau = s == 0 ? a.tau0 : (s == 1 ? a.tau1 : a.tau2);
    const int nunits = a.Z >> 5;
    const uint32_t col0 = (uint32_t)tile * BN + cl * 8;
    const unsigned char* wp = a.wq + (size_t)tile * (BN / 2) + cl * 4;
    uint32_t* list = lists[wave];
    float total[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) total[k] = 0.0f;
    // unit u belongs to slice u % split, and inside the slice to wave (u / split) % 16
    const int ustride = split * WAVES;
    for (int u0 = slice + split * wave; u0 < nunits; u0 += ustride * UP) {
        // ---- 1. activations and group parameters of the pass ------------------------------------------------------
        uint32_t xb[UP];
        u32x4 sz0[UP], sz1[UP];
        bool live[UP];
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            const int u = u0 + i * ustride;
            live[i] = u < nunits;  // wave-uniform
            const int uu = live[i] ? u : u0;
            xb[i] = a.x[(uu << 5) + (lane & 31)];
            const u32x4* szp = reinterpret_cast<const u32x4*>(a.sz + ((size_t)((uu << 5) / a.G) * a.N + col0) * 2);
            sz0[i] = szp[0];
            sz1[i] = szp[1];
        }
        // ---- 2. ballots -> (row in unit : 16 | x bits : 16) pairs in the wave's list, ascending -------------------------
        int off[UP + 1];
        off[0] = 0;
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            const float v = bits_to_float(xb[i], BF16);
            const uint32_t mask = live[i] ? (uint32_t)__ballot(keep_rule(v, tau) || (v != v)) : 0u;  // lanes 32..63 mirror 0..31
            if (lane < 32 && ((mask >> lane) & 1u))
                list[off[i] + __builtin_amdgcn_mbcnt_lo(mask, 0u)] = ((uint32_t)lane << 16) | xb[i];
            off[i + 1] = off[i] + __popc(mask);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- 3. every row load of the pass (up to 8 steps of 4 rows per unit) ----------------------------------------------
        uint32_t d[UP][8];
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            const int row0 = (live[i] ? u0 + i * ustride : u0) << 5;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                d[i][r] = 0u;
                if (off[i] + 4 * r < off[i + 1]) {  // wave-uniform: this step has rows
                    const int e = off[i] + 4 * r + rs;
                    if (e < off[i + 1])
                        d[i][r] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(wp + (size_t)(row0 + (int)(list[e] >> 16)) * a.ldb));
                }
            }
        }
        // ---- 4. arithmetic, unit by unit; scale / zero once per (unit, column) -------------------------------------------
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            float A[8], X = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = 0.0f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (off[i] + 4 * r < off[i + 1]) {
                    const int e = off[i] + 4 * r + rs;
                    const float xr = e < off[i + 1] ? bits_to_float(list[e] & 0xFFFFu, BF16) : 0.0f;  // a lane without a row adds 0
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const f16x2 hq = __builtin_bit_cast(f16x2, ((d[i][r] >> (4 * jj)) & 0x000F000Fu) | 0x64006400u);
                        A[jj] = fmaf((float)hq.x, xr, A[jj]);
                        A[jj + 4] = fmaf((float)hq.y, xr, A[jj + 4]);
                    }
                    X += xr;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t pr = k < 4 ? sz0[i][k] : sz1[i][k - 4];  // bf16 pair: scale (low half), zero (high half)
                const float sc = __uint_as_float(pr << 16), zr = __uint_as_float(pr & 0xFFFF0000u);
                total[k] += sc * (A[k] - 1032.0f * X) + zr * X;  // (q - 8) = (1024 + q) - 1032; a dead unit has X = A = 0
            }
        }
        __builtin_amdgcn_wave_barrier();  // the list is rewritten by the next pass
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
    const int ntiles = N / 128, nunits = Z / 32;
    int split = dc->num_cu / ntiles;
    if (split > 8) split = 8;
    if (split * 16 > nunits) split = nunits / 16;  // every wave of every slice owns at least one 32-row unit
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
