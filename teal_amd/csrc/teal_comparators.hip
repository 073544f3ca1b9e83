// teal_comparators.hip — benchmark comparators of the micro-benchmark (scripts/benchmark_gemv.py), NOT on the decode path.
//
// The reference's kernel benchmark plots TEAL's sparse GEMV against the Deja Vu gather kernel
// (scripts/benchmark_gemv.py:32-107 gather_transposed_gemv_flag_atomicadd_kernel, :170-172 deja_vu_gemv): a boolean flag per
// input feature is computed by separate launches (x.abs() > s / 2), the output is zeroed, and every program loads a
// BLOCK_M x BLOCK_N block of W^T masked by its rows' flags and adds its fp32 partial column sums into Y with atomics.
// This is that METHOD restated for CDNA4 so that the MI355X plot has the reference's four lines — precomputed flags instead
// of in-kernel compaction, a (row block, column tile) grid instead of row lists, fp32 atomics into a zeroed fp32 output —
// with the memory idiom of this repo (16-byte non-temporal loads, wave64).  Three launches per GEMV, like the original.
#include "teal_common.h"

namespace teal {

template <bool BF16>
__global__ __launch_bounds__(256) void cmp_flags_kernel(const uint16_t* __restrict__ x, const int Z, const float tau,
                                                        unsigned char* __restrict__ flags) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < Z) flags[m] = keep_rule(bits_to_float(x[m], BF16), tau) ? 1 : 0;
}

// grid (Z / 64 row blocks, N / 512 column tiles), 256 threads: wave w takes rows 16 w .. 16 w + 15 of the block, a lane 8
// columns (16 bytes); rows whose flag is 0 are skipped (a masked load); the four waves' partials are summed through LDS and
// added to y with fp32 atomics (one per column and row block)
template <bool BF16>
__global__ __launch_bounds__(256) void cmp_flag_gemv_atomic_kernel(const uint16_t* __restrict__ x, const unsigned char* __restrict__ flags,
                                                                   const uint16_t* __restrict__ wT, const int ld, float* __restrict__ y,
                                                                   const int Z, const int N) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y * 512 + lane * 8;
    const int r0 = blockIdx.x * 64 + wave * 16;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    if (col < N) {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int m = r0 + i;
            if (m < Z && flags[m]) {  // wave-uniform
                const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wT + (size_t)m * ld + col));
                const float xv = bits_to_float(x[m], BF16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * j] = fmaf(bits_to_float(w[j] & 0xFFFFu, BF16), xv, acc[2 * j]);
                    acc[2 * j + 1] = fmaf(bits_to_float(w[j] >> 16, BF16), xv, acc[2 * j + 1]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = acc[j];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int n = blockIdx.y * 512 + c;
        if (n < N) atomicAdd(&y[n], (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
    }
}

}  // namespace teal

using namespace teal;

extern "C" int teal_cmp_flag_gemv(const void* x, const void* wT, int ld, float* y32, unsigned char* flags, float tau, int Z, int N,
                                  int dtype, void* stream) {
    if (!x || !wT || !y32 || !flags || Z <= 0 || N <= 0 || ld < N) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if ((N & 7) || (ld & 7)) return TEAL_ERR_SHAPE;
    if (!aligned16(wT)) return TEAL_ERR_ALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto* xp = reinterpret_cast<const uint16_t*>(x);
    auto* wp = reinterpret_cast<const uint16_t*>(wT);
    const dim3 g1((Z + 255) / 256), g2((Z + 63) / 64, (N + 511) / 512);
    if (hipMemsetAsync(y32, 0, (size_t)N * sizeof(float), st) != hipSuccess) return TEAL_ERR_LAUNCH;  // init_to_zero("Y")
    if (dtype == TEAL_BF16) {
        hipLaunchKernelGGL((cmp_flags_kernel<true>), g1, dim3(256), 0, st, xp, Z, tau, flags);
        hipLaunchKernelGGL((cmp_flag_gemv_atomic_kernel<true>), g2, dim3(256), 0, st, xp, flags, wp, ld, y32, Z, N);
    } else {
        hipLaunchKernelGGL((cmp_flags_kernel<false>), g1, dim3(256), 0, st, xp, Z, tau, flags);
        hipLaunchKernelGGL((cmp_flag_gemv_atomic_kernel<false>), g2, dim3(256), 0, st, xp, flags, wp, ld, y32, Z, N);
    }
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}
