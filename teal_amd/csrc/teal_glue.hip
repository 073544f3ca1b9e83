// teal_glue.hip — the two element-wise steps of a decode layer as launches of their own, for weight formats whose GEMV kernel
// takes a plain activation vector (the int4 group-quantised kernel, teal_gemv_int4.hip): the 16-bit and int8 engines fold
// them into the GEMV launches as producers (teal_gemv_fast.h MODE 1 / MODE 2) and never launch these.
//
//   teal_resid_rmsnorm   h = resid + add;  x = RMSNorm(h) * w          gpt-fast/model.py:158-161, 289-291
//   teal_silu_mul        h = silu(gate) * up                            gpt-fast/model.py:258-259
//
// Same roundings as the fused producers (h rounded to dtype, x = round(round(h * rstd) * w); silu rounded, product rounded),
// so an int4 engine step and the 16-bit engine differ only by their weights.
#include "teal_common.h"

namespace teal {

// one workgroup of 1024 threads (Z <= 16384: up to 16 elements per thread, held in registers)
template <bool BF16>
__global__ __launch_bounds__(1024) void resid_rmsnorm_kernel(const uint16_t* __restrict__ resid_in, const int* __restrict__ row_index,
                                                             const uint16_t* __restrict__ add, const uint16_t* __restrict__ norm_w,
                                                             const float eps, uint16_t* __restrict__ resid_out,
                                                             uint16_t* __restrict__ x_out, const int Z) {
    __shared__ float wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t* r = resid_in + (row_index ? (size_t)row_index[0] * (size_t)Z : 0);
    float hv[16];
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int m = tid + k * 1024;
        float h = 0.0f;
        if (m < Z) {
            h = bits_to_float(r[m], BF16);
            if (add) h = bits_to_float(float_to_bits<BF16>(h + bits_to_float(add[m], BF16)), BF16);
        }
        hv[k] = h;
        ss = fmaf(h, h, ss);
    }
    ss = wave_sum_f(ss);
    if (lane == 0) wsum[wave] = ss;
    __syncthreads();
    float tot = lane < 16 ? wsum[lane] : 0.0f;
    tot = wave_sum_f(tot);
    const float rstd = rsqrtf(tot / (float)Z + eps);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int m = tid + k * 1024;
        if (m < Z) {
            const float xn = bits_to_float(float_to_bits<BF16>(hv[k] * rstd), BF16);
            x_out[m] = float_to_bits<BF16>(xn * bits_to_float(norm_w[m], BF16));
            if (resid_out) resid_out[m] = float_to_bits<BF16>(hv[k]);
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ gate, const uint16_t* __restrict__ up,
                                                       uint16_t* __restrict__ h, const int Z) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= Z) return;
    const float g = bits_to_float(gate[m], BF16);
    const float sl = bits_to_float(float_to_bits<BF16>(g / (1.0f + expf(-g))), BF16);
    h[m] = float_to_bits<BF16>(sl * bits_to_float(up[m], BF16));
}

}  // namespace teal

using namespace teal;

extern "C" int teal_resid_rmsnorm(const void* resid_in, const int32_t* row_index, const void* add, const void* norm_weight, float eps,
                                  void* resid_out, void* x_out, int Z, int dtype, void* stream) {
    if (!resid_in || !norm_weight || !x_out || Z <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    if (Z > 16384) return TEAL_ERR_SHAPE;
    if (resid_out == resid_in && !row_index) return TEAL_ERR_ARG;  // must ping-pong (every thread reads before any writes otherwise)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define TEAL_RN(BF) hipLaunchKernelGGL((resid_rmsnorm_kernel<BF>), dim3(1), dim3(1024), 0, st, reinterpret_cast<const uint16_t*>(resid_in), row_index, \
                                       reinterpret_cast<const uint16_t*>(add), reinterpret_cast<const uint16_t*>(norm_weight), eps,                  \
                                       reinterpret_cast<uint16_t*>(resid_out), reinterpret_cast<uint16_t*>(x_out), Z)
    if (dtype == TEAL_BF16) TEAL_RN(true); else TEAL_RN(false);
#undef TEAL_RN
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}

extern "C" int teal_silu_mul(const void* gate, const void* up, void* h, int Z, int dtype, void* stream) {
    if (!gate || !up || !h || Z <= 0) return TEAL_ERR_ARG;
    if (dtype != TEAL_F16 && dtype != TEAL_BF16) return TEAL_ERR_DTYPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((Z + 255) / 256), block(256);
    if (dtype == TEAL_BF16)
        hipLaunchKernelGGL((silu_mul_kernel<true>), grid, block, 0, st, reinterpret_cast<const uint16_t*>(gate),
                           reinterpret_cast<const uint16_t*>(up), reinterpret_cast<uint16_t*>(h), Z);
    else
        hipLaunchKernelGGL((silu_mul_kernel<false>), grid, block, 0, st, reinterpret_cast<const uint16_t*>(gate),
                           reinterpret_cast<const uint16_t*>(up), reinterpret_cast<uint16_t*>(h), Z);
    return hipGetLastError() == hipSuccess ? TEAL_OK : TEAL_ERR_LAUNCH;
}
