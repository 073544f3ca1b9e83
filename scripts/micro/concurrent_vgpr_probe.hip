// concurrent_vgpr_probe — does a trivially deterministic kernel (1024-thread workgroups, ~60 live VGPRs, DPP / ds_swizzle /
// ds_bpermute cross-lane steps, an LDS exchange, one write per thread; NO global loads but its own arguments) return the same
// bits launch after launch while a SECOND PROCESS uses the same GPU and both churn hipMalloc / hipFree?  Round 6: the fused
// decode step showed single-VGPR-sized differences (one accumulator of one wave: 16 columns, stride 8, of one tile) in ~3 % of
// first steps with two processes on the GPU and none with one (scripts/micro/first_call_probe.py); this probe separates the
// platform from the kernels.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/concurrent_vgpr_probe.hip -o scripts/micro/concurrent_vgpr_probe
//   ./concurrent_vgpr_probe [launches] [churn_every] [tag]      (run two at once)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NACC = 48;

__global__ __launch_bounds__(1024) void regs_kernel(float* __restrict__ out, const int iters, const float c1, const float c2) {
    __shared__ float red[16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 1.0f + 0.001f * (float)((tid * 7 + j * 13 + blockIdx.x) & 1023);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = fmaf(acc[j], c1, c2 + 1e-6f * (float)j);
    }
    // cross-lane steps as the GEMV epilogue has them: DPP row rotate, ds_swizzle swap, ds_bpermute shuffle
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        float v = acc[j];
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));  // row_ror:8
        v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));                      // SWAP,16
        v += __shfl_xor(v, 32);
        acc[j] = v;
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j] * (1.0f + 0.01f * (float)j);
    red[wave * 64 + lane] = s;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < 16; ++w) t += red[w * 64 + lane] * (1.0f + 0.1f * (float)((w + wave) & 15));
    out[(size_t)blockIdx.x * 1024 + tid] = t + s;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20000, churn = argc > 2 ? atoi(argv[2]) : 200;
    const char* tag = argc > 3 ? argv[3] : "p";
    const int wgs = 256, n = wgs * 1024, iters = 400;  // ~15-20 us per launch
    float* out; CK(hipMalloc(&out, (size_t)n * 4));
    std::vector<float> ref(n), cur(n);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipLaunchKernelGGL(regs_kernel, dim3(wgs), dim3(1024), 0, st, out, iters, 0.9991f, 0.0013f);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(ref.data(), out, (size_t)n * 4, hipMemcpyDeviceToHost));
    long bad_launches = 0, bad_words = 0;
    for (int i = 0; i < launches; ++i) {
        CK(hipMemsetAsync(out, 0xFF, (size_t)n * 4, st));
        hipLaunchKernelGGL(regs_kernel, dim3(wgs), dim3(1024), 0, st, out, iters, 0.9991f, 0.0013f);
        CK(hipMemcpyAsync(cur.data(), out, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        if (memcmp(cur.data(), ref.data(), (size_t)n * 4) != 0) {
            long nb = 0, first = -1, last = -1;
            for (int k = 0; k < n; ++k) if (memcmp(&cur[k], &ref[k], 4)) { ++nb; if (first < 0) first = k; last = k; }
            ++bad_launches; bad_words += nb;
            if (bad_launches <= 8)
                printf("[%s] launch %d: %ld of %d words differ; first %ld (workgroup %ld, wave %ld, lane %ld) last %ld; ref %.9g now %.9g\n", tag, i, nb, n,
                       first, first / 1024, (first % 1024) / 64, first % 64, last, ref[first], cur[first]);
        }
        if (churn > 0 && i % churn == churn - 1) {  // allocation churn, as an engine being rebuilt does
            void* p[4];
            for (int k = 0; k < 4; ++k) { CK(hipMalloc(&p[k], (size_t)(64 + 32 * k) << 20)); CK(hipMemsetAsync(p[k], k, (size_t)(64 + 32 * k) << 20, st)); }
            CK(hipStreamSynchronize(st));
            for (int k = 0; k < 4; ++k) CK(hipFree(p[k]));
        }
    }
    printf("[%s] %d launches: %ld launches differed (%ld words)\n", tag, launches, bad_launches, bad_words);
    return bad_launches ? 3 : 0;
}
