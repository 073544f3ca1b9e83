// concurrent_stream_probe — pure HIP, none of the product's code: does a memory-streaming GEMV-like kernel (16-byte row-segment
// loads of an fp16 matrix, 8 fp32 accumulators per lane, cross-lane reduce, one store per column) return the same bits launch
// after launch while ANOTHER PROCESS runs dense prompt passes (rocBLAS / hipBLASLt GEMMs) on the same GPU?  Round 6: the fused
// decode step's outputs differed in whole-accumulator patterns (16 or 8 columns at stride 8 of one tile) only while a second
// process ran its dense prefill next to it (scripts/micro/concurrency_determinism_probe.py --noise prefill).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/concurrent_stream_probe.hip -o scripts/micro/concurrent_stream_probe
//   ./concurrent_stream_probe <mode> [launches] [tag]     mode 0: non-temporal loads, 1: plain loads, 2: no loads (ALU only), 3: as 0 with 64 VGPRs allocated, 4: with 104,
//                                                         5: as 0 with the accumulators advanced by v_pk_fma_f32 (packed fp32, what the product's inner loops used until round 6),
//                                                         6: no victim — run the MFMA aggressor (small workgroups of v_mfma_f32_16x16x16_f16 loops) until killed,
//                                                         7: mode 5 with that aggressor on a second stream of the SAME process,
//                                                         8 / 9: mode 5 with 104 / 124 VGPRs allocated (the product kernels' occupancy: one 16-wave workgroup owns the CU's registers)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int Z = 4096, N = 11008, BN = 128, LPR = 16, RPW = 4;  // 86 tiles x 3 slices of rows = 258 workgroups of 16 waves

template <int MODE, int NV = 0>
__global__ __launch_bounds__(1024) void stream_kernel(const uint16_t* __restrict__ W, const float* __restrict__ x, float* __restrict__ out,
                                                      const int rows_per_slice) {
    __shared__ float red[16 * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, slice = blockIdx.y;
    const int g = lane / LPR, cl = lane % LPR;
    const char* wp = reinterpret_cast<const char*>(W) + ((size_t)tile * BN + cl * 8) * 2;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    // NV: the register allocation of the wave (the kernel itself needs 28): a clobbered high register raises it to 64 / 104
    if constexpr (NV == 64) asm volatile("v_mov_b32 v61, 0" ::: "v61");
    if constexpr (NV == 104) asm volatile("v_mov_b32 v100, 0" ::: "v100");
    if constexpr (NV == 124) asm volatile("v_mov_b32 v123, 0" ::: "v123");
    const int r0 = slice * rows_per_slice, rend = min(r0 + rows_per_slice, Z);
    for (int r = r0 + wave * RPW + g; r < rend; r += 16 * RPW * 4) {
        u32x4 w[4];
        float xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = min(r + u * 16 * RPW, rend - 1);
            xv[u] = (r + u * 16 * RPW < rend) ? x[rr] : 0.0f;
            if (MODE == 0 || MODE == 5) w[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (size_t)rr * N * 2));
            else if (MODE == 1) w[u] = *reinterpret_cast<const u32x4*>(wp + (size_t)rr * N * 2);
            else w[u] = u32x4{0x3c003c00u + (uint32_t)rr, 0x38003800u, 0x34003400u + (uint32_t)cl, 0x30003000u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f16x2 h = __builtin_bit_cast(f16x2, w[u][j]);
                if constexpr (MODE == 5) {  // two fp32 results per instruction: v_pk_fma_f32 (check the ISA: llvm-objdump -d)
                    f32x2 a2 = {acc[2 * j], acc[2 * j + 1]};
                    a2 = __builtin_elementwise_fma((f32x2){(float)h.x, (float)h.y}, (f32x2){xv[u], xv[u]}, a2);
                    acc[2 * j] = a2.x;
                    acc[2 * j + 1] = a2.y;
                } else {
                    acc[2 * j] = fmaf((float)h.x, xv[u], acc[2 * j]);
                    acc[2 * j + 1] = fmaf((float)h.y, xv[u], acc[2 * j + 1]);
                }
            }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = acc[j];
        v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));  // lane ^ 16
        v += __shfl_xor(v, 32);
        acc[j] = v;
    }
    if (lane < LPR) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave * BN + lane * 8 + j] = acc[j];
    }
    __syncthreads();
    if (tid < BN) {
        float s = 0.0f;
        for (int wv = 0; wv < 16; ++wv) s += red[wv * BN + tid];
        out[((size_t)tile * BN + tid) * 4 + slice] = s;
    }
}

// The aggressor: one-wave workgroups (few registers, no LDS: they fit next to anything) spinning on the matrix core.
__global__ __launch_bounds__(64) void mfma_kernel(float* __restrict__ sink, const int iters) {
    f16x4 a = {(_Float16)(threadIdx.x * 0.001f), (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.0f};
    f16x4 b = {(_Float16)1.0f, (_Float16)(blockIdx.x * 0.001f), (_Float16)0.125f, (_Float16)2.0f};
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16f16(b, a, c, 0, 0, 0);
    }
    if (c[0] == 12345.678f) sink[threadIdx.x] = c[1] + c[2] + c[3];
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, launches = argc > 2 ? atoi(argv[2]) : 20000;
    const char* tag = argc > 3 ? argv[3] : "p";
    const int slices = 3, rps = (Z + slices - 1) / slices;
    uint16_t* W; float *x, *out;
    CK(hipMalloc(&W, (size_t)Z * N * 2)); CK(hipMalloc(&x, Z * 4)); CK(hipMalloc(&out, (size_t)N * 4 * 4));
    {
        std::vector<uint16_t> hw((size_t)Z * N); std::vector<float> hx(Z);
        uint32_t s = 12345u;
        for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x2000u + ((s >> 9) & 0x0FFFu) + ((s >> 3) & 0x8000u)); }  // |w| in [2^-7, 2^-5)
        for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) / 500.0f; }
        CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), Z * 4, hipMemcpyHostToDevice));
    }
    const size_t n = (size_t)N * 4;
    std::vector<float> ref(n), cur(n), again(n);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipStream_t st2; CK(hipStreamCreate(&st2));
    float* sink; CK(hipMalloc(&sink, 4096));
    const int agg_blocks = argc > 4 ? atoi(argv[4]) : 2048, agg_iters = argc > 5 ? atoi(argv[5]) : 4000;
    if (mode == 6) {  // aggressor only, until killed
        printf("[%s] MFMA aggressor: %d one-wave workgroups x %d x 2 v_mfma_f32_16x16x16_f16 per launch, forever\n", tag, agg_blocks, agg_iters); fflush(stdout);
        for (;;) {
            for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(mfma_kernel, dim3(agg_blocks), dim3(64), 0, st2, sink, agg_iters);
            CK(hipStreamSynchronize(st2));
        }
    }
    auto launch = [&]() {
        if (mode == 7) hipLaunchKernelGGL(mfma_kernel, dim3(agg_blocks), dim3(64), 0, st2, sink, agg_iters);
        const dim3 grid(N / BN, slices), block(1024);
        if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, grid, block, 0, st, W, x, out, rps);
        else if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, grid, block, 0, st, W, x, out, rps);
        else if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, grid, block, 0, st, W, x, out, rps);
        else if (mode == 3) hipLaunchKernelGGL((stream_kernel<0, 64>), grid, block, 0, st, W, x, out, rps);
        else if (mode == 5 || mode == 7) hipLaunchKernelGGL(stream_kernel<5>, grid, block, 0, st, W, x, out, rps);
        else if (mode == 8) hipLaunchKernelGGL((stream_kernel<5, 104>), grid, block, 0, st, W, x, out, rps);   // packed, 104 VGPRs allocated
        else if (mode == 9) hipLaunchKernelGGL((stream_kernel<5, 124>), grid, block, 0, st, W, x, out, rps);   // packed, 124 VGPRs allocated: one workgroup fills a CU's registers
        else hipLaunchKernelGGL((stream_kernel<0, 104>), grid, block, 0, st, W, x, out, rps);
    };
    CK(hipMemsetAsync(out, 0, n * 4, st));
    launch();
    CK(hipMemcpyAsync(ref.data(), out, n * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    long bad_launches = 0, bad_words = 0;
    for (int i = 0; i < launches; ++i) {
        launch();
        CK(hipMemcpyAsync(cur.data(), out, n * 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        if (memcmp(cur.data(), ref.data(), n * 4) != 0) {
            long nb = 0;
            std::vector<long> idx;
            for (size_t k = 0; k < n; ++k) if (memcmp(&cur[k], &ref[k], 4)) { ++nb; if (idx.size() < 12) idx.push_back((long)k); }
            ++bad_launches; bad_words += nb;
            // persistence: the same launch again, five times
            int same_again = 0, clean_again = 0;
            for (int t = 0; t < 5; ++t) {
                launch();
                CK(hipMemcpyAsync(again.data(), out, n * 4, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                same_again += memcmp(again.data(), cur.data(), n * 4) == 0;
                clean_again += memcmp(again.data(), ref.data(), n * 4) == 0;
            }
            if (bad_launches <= 10) {
                printf("[%s mode %d] launch %d: %ld of %zu words differ; first (column:slice)", tag, mode, i, nb, n);
                for (long k : idx) printf(" %ld:%ld", k / 4, k % 4);
                printf("; ref %.7g now %.7g; of 5 relaunches %d identical to the bad one, %d clean\n", ref[idx[0]], cur[idx[0]], same_again, clean_again);
                fflush(stdout);
            }
        }
    }
    printf("[%s mode %d] %d launches: %ld launches differed (%ld words)\n", tag, mode, launches, bad_launches, bad_words);
    return 0;
}
