// valu_rate_probe.hip — issue rate of the vector instructions the int4 kernel's inner loop is made of, on gfx950:
// one 1024-thread workgroup per CU (16 waves = 4 per SIMD, like the kernel), each wave runs ITER x 16 independent
// instructions of one kind; cycles per wave-instruction per SIMD = clocks * 1 / (4 waves * ITER * 16).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;
template <int KIND>
__global__ __launch_bounds__(1024) void probe(float* out, uint32_t seed, unsigned long long* clk) {
    float acc[16];
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc[k] = (float)k; w[k] = seed * (k + 1) + threadIdx.x; }
    const uint32_t x2 = 0x3C003C00u + (seed & 1);
    float xf = 1.0f + (float)(seed & 1);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (KIND == 0) acc[k] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w[k]), __builtin_bit_cast(h2, x2), acc[k], false);
            else if constexpr (KIND == 1) acc[k] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, w[k]), __builtin_bit_cast(b2, x2), acc[k], false);
            else if constexpr (KIND == 2) acc[k] = fmaf(acc[k], xf, 1.0f);
            else if constexpr (KIND == 3) acc[k] = fmaf((float)__builtin_bit_cast(h2, w[k]).x, xf, acc[k]);  // v_fma_mix_f32
            else if constexpr (KIND == 4) { w[k] = (w[k] & 0x000F000Fu) | (x2 + k); }                     // and + or (or v_and_or)
            else if constexpr (KIND == 5) { w[k] = (w[k] >> 4) ^ x2; }
        }
        if constexpr (KIND == 6) {  // packed fp32 fma: 8 instructions of 2 lanes-worth each
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                f2 a = {acc[k], acc[k + 1]};
                a = a * f2{xf, xf} + f2{1.0f, 1.0f};
                acc[k] = a.x; acc[k + 1] = a.y;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += acc[k] + (float)w[k];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int per_iter) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<256, 1024>>>(out, 3, clk);
    hipEventRecord(e0);
    probe<KIND><<<256, 1024>>>(out, 5, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i]; c /= 256;
    const double winstr = 4.0 * ITER * per_iter;  // wave-instructions per SIMD
    printf("%-28s %8.1f us   %.2f shader clocks per wave-instruction per SIMD  (%.2f ns)\n", name, ms * 1e3, c / winstr, ms * 1e6 / winstr);
    hipFree(out); hipFree(clk);
}
int main() {
    run<0>("v_dot2c_f32_f16", 16);
    run<1>("v_dot2c_f32_bf16", 16);
    run<2>("v_fmac_f32", 16);
    run<3>("v_fma_mix_f32 (f16 x f32)", 16);
    run<4>("v_and + v_or (2 instr)", 32);
    run<5>("v_lshr + v_xor (2 instr)", 32);
    run<6>("v_pk_fma_f32", 8);
    return 0;
}
