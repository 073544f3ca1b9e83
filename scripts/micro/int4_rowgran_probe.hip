// int4_rowgran_probe — does a ROW-GRANULAR int4 image with 256-column tiles stream faster than the product's row-PAIR image
// with 128-column tiles?  (round-3 verdict, item 3: "a kept row is then fetched alone at 50 %, not at 75 %".)
//
// The probe kernel keeps the product kernel's organisation (teal_amd/csrc/teal_gemv_int4.hip: 32-row units, a unit per 16-lane
// group, group parameters applied once per (unit, column), every load of a pass issued before the first is consumed, fp32
// partial slabs out) and changes exactly what the verdict names:
//   image   W^T by ROWS, [Z][N / 2] bytes (+ 128 of padding): nibble j of dword d of a row = column 8 d + j
//   tile    256 columns = one 128-byte segment per kept row, 16 lanes x 8 bytes (a lane: 16 columns, 16 accumulators)
//   arithmetic  (1024 + q) / (1024 + 16 q) under an fp16 exponent as in the product, but one row per word: v_fma_mix_f32 per
//           multiply-add instead of one v_dot2_f32_f16 per two
// and times it against the product launch (teal_fused_gemv, weight_bits = 4, plain x, slab output) on the four Llama-2-7B
// shapes at 50 % activation sparsity, weights rotating over 8 images.  Checked against a naive fp64 reference kernel.
// Benchmark utility, not product code.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include scripts/micro/int4_rowgran_probe.hip -DTEAL_DIAGNOSTICS -L teal_amd -lteal_hip_diag \
//         -Wl,-rpath,'$ORIGIN/../../teal_amd' -o scripts/micro/int4_rowgran_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "teal_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define TK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "teal error %d (%s) at %s:%d\n", r_, teal_strerror(r_), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }
__global__ void fill_bytes(unsigned char* p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (unsigned char)(mix((uint32_t)i * 0x9E3779B9u + seed) >> 13);
}
__global__ void fill_x(uint16_t* p, int n, uint32_t seed) {  // U(-1, 1) halves
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = (float)(mix((uint32_t)i * 0x9E3779B9u + seed) >> 8) * (2.0f / 16777216.0f) - 1.0f; _Float16 h = (_Float16)v; p[i] = __builtin_bit_cast(uint16_t, h); }
}
__global__ void fill_sz(uint16_t* p, size_t n, uint32_t seed) {  // bf16 (scale ~ 0.01 .. 0.02, zero ~ +-0.05), interleaved
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float u = (float)(mix((uint32_t)i * 0x9E3779B9u + seed) >> 8) * (1.0f / 16777216.0f);
        const float v = (i & 1) ? (u - 0.5f) * 0.1f : 0.01f + 0.01f * u;
        p[i] = (uint16_t)(__float_as_uint(v) >> 16);
    }
}

// reference: one thread per column, fp64
__global__ void ref_kernel(const uint16_t* x, const unsigned char* wq, const uint16_t* sz, double* y, int Z, int N, int ldb, float tau) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double acc = 0.0;
    for (int m = 0; m < Z; ++m) {
        const float xv = (float)__builtin_bit_cast(_Float16, x[m]);
        if (!(fabsf(xv) > tau)) continue;
        const unsigned char b = wq[(size_t)m * ldb + (n >> 1)];
        const int q = (n & 1) ? (b >> 4) : (b & 15);
        const uint16_t* pz = sz + ((size_t)(m >> 5) * N + n) * 2;
        const float sc = __uint_as_float((uint32_t)pz[0] << 16), zr = __uint_as_float((uint32_t)pz[1] << 16);
        acc += (double)xv * ((double)(q - 8) * (double)sc + (double)zr);
    }
    y[n] = acc;
}

__device__ __forceinline__ float xadd(float v, int off) { return v + __shfl_xor(v, off); }

// ROUNDS_OF: loads a lane has in flight per round (16: two rounds for a unit with more than 16 kept rows)
__global__ __launch_bounds__(1024) void i4rg_kernel(const uint16_t* __restrict__ x, const unsigned char* __restrict__ wq, const uint16_t* __restrict__ sz,
                                                     float* __restrict__ ws, const int Z, const int N, const int ldb, const float tau, const int ws_stride) {
    constexpr int WAVES = 16, BN = 256, UP = 4;
    __shared__ float red[WAVES * BN];
    __shared__ __align__(16) uint8_t list_i[WAVES][UP * 32];
    __shared__ __align__(16) uint16_t list_x[WAVES][UP * 32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, slice = blockIdx.y, split = gridDim.y;
    const int rs = lane >> 4, cl = lane & 15;
    const int nunits = Z >> 5;
    const unsigned char* wtile = wq + (size_t)tile * (BN / 2) + (size_t)cl * 8;
    const uint32_t uldb = (uint32_t)ldb;
    const uint16_t* szb = sz + ((size_t)tile * BN + (size_t)cl * 16) * 2;
    uint8_t* li = list_i[wave];
    uint16_t* lx = list_x[wave];
    float total[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) total[k] = 0.0f;
    const int ustride = split * WAVES;
    for (int u0 = slice + split * wave; u0 < nunits; u0 += ustride * UP) {
        int cnt[UP];
        // ---- activations, ballots, the wave's lists (a unit's kept ROWS, ascending) ----
        uint32_t xb[UP];
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            const int u = u0 + i * ustride;
            const int uu = u < nunits ? u : u0;
            xb[i] = x[(uint32_t)(uu << 5) + (lane & 31)];
        }
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            const bool live = u0 + i * ustride < nunits;
            const float v = (float)__builtin_bit_cast(_Float16, (uint16_t)xb[i]);
            const bool keep = fabsf(v) > tau || v != v;
            const uint32_t mask = live ? (uint32_t)__ballot(keep) : 0u;
            if (lane < 32 && ((mask >> lane) & 1u)) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_lo(mask, 0u);
                li[i * 32 + rank] = (uint8_t)lane;
                lx[i * 32 + rank] = (uint16_t)xb[i];
            }
            cnt[i] = __popc(mask);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- lane group rs takes unit rs of the pass ----
        const int cntv = rs == 0 ? cnt[0] : (rs == 1 ? cnt[1] : (rs == 2 ? cnt[2] : cnt[3]));
        const int maxcnt = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
        const int uv = min(u0 + rs * ustride, nunits - 1);
        const u32x4* szp = reinterpret_cast<const u32x4*>(szb + (size_t)uv * N * 2);  // g32: one parameter row per unit
        const u32x4 p0 = szp[0], p1 = szp[1], p2 = szp[2], p3 = szp[3];
        const uint32_t row0 = (uint32_t)(uv << 5);
        float A[16], X = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) A[k] = 0.0f;
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 16) {
            if (r0 < maxcnt) {  // wave-uniform
                u32x2 d[16];
                const u32x4 pidx = *reinterpret_cast<const u32x4*>(li + rs * 32 + r0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    d[r] = u32x2{0u, 0u};
                    if (r0 + r < maxcnt) {
                        if (r0 + r < cntv)
                            d[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wtile + (size_t)(row0 + ((pidx[r >> 2] >> (8 * (r & 3))) & 0xFFu)) * uldb));
                    }
                }
                const u32x4 xa = *reinterpret_cast<const u32x4*>(lx + rs * 32 + r0), xc = *reinterpret_cast<const u32x4*>(lx + rs * 32 + r0 + 8);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r0 + r < maxcnt) {
                        const uint32_t xw = r < 8 ? xa[r >> 1] : xc[(r - 8) >> 1];
                        const uint16_t xh = (r & 1) ? (uint16_t)(xw >> 16) : (uint16_t)xw;
                        const float xf = (r0 + r < cntv) ? (float)__builtin_bit_cast(_Float16, xh) : 0.0f;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint32_t w = d[r][h], w8 = w >> 8;
                            const f16x2 t0 = __builtin_bit_cast(f16x2, (w & 0x000F000Fu) | 0x64006400u);
                            const f16x2 t1 = __builtin_bit_cast(f16x2, (w & 0x00F000F0u) | 0x64006400u);
                            const f16x2 t2 = __builtin_bit_cast(f16x2, (w8 & 0x000F000Fu) | 0x64006400u);
                            const f16x2 t3 = __builtin_bit_cast(f16x2, (w8 & 0x00F000F0u) | 0x64006400u);
                            A[8 * h + 0] = fmaf((float)t0.x, xf, A[8 * h + 0]); A[8 * h + 4] = fmaf((float)t0.y, xf, A[8 * h + 4]);
                            A[8 * h + 1] = fmaf((float)t1.x, xf, A[8 * h + 1]); A[8 * h + 5] = fmaf((float)t1.y, xf, A[8 * h + 5]);
                            A[8 * h + 2] = fmaf((float)t2.x, xf, A[8 * h + 2]); A[8 * h + 6] = fmaf((float)t2.y, xf, A[8 * h + 6]);
                            A[8 * h + 3] = fmaf((float)t3.x, xf, A[8 * h + 3]); A[8 * h + 7] = fmaf((float)t3.y, xf, A[8 * h + 7]);
                        }
                        X += xf;
                    }
                }
            }
        }
        // ---- scale / zero once per (unit, column) ----
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t pr = k < 4 ? p0[k] : (k < 8 ? p1[k - 4] : (k < 12 ? p2[k - 8] : p3[k - 12]));
            const float sc = __uint_as_float(pr << 16), zr = __uint_as_float(pr & 0xFFFF0000u);
            const float t = (k & 1) ? fmaf(A[k], 0.0625f, -72.0f * X) : A[k] - 1032.0f * X;
            total[k] += fmaf(sc, t, zr * X);
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) total[k] = xadd(xadd(total[k], 16), 32);
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) red[wave * BN + lane * 16 + k] = total[k];
    }
    __syncthreads();
    if (tid < BN) {
        float sum = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sum += red[w * BN + tid];
        ws[((size_t)tile * BN + tid) * ws_stride + slice] = sum;
    }
}

int main(int argc, char** argv) {
    const int ncu = teal_init();
    if (ncu <= 0) { fprintf(stderr, "no device\n"); return 1; }
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Shape { const char* name; int Z, N; };
    const Shape shapes[4] = {{"gate|up 4096 -> 2 x 11008", 4096, 22016}, {"qkv 4096 -> 12288", 4096, 12288}, {"down 11008 -> 4096", 11008, 4096}, {"wo 4096 -> 4096", 4096, 4096}};
    const float tau = 0.5f;  // x = U(-1, 1): half of the rows kept
    const int NIMG = 8;
    printf("int4-g32, fp16 activations, 50 %% activation sparsity, %d CUs; us per launch incl. the launch boundary (hipGraph of 32 launches rotating over %d images)\n", ncu, NIMG);
    for (const Shape& S : shapes) {
        const int Z = S.Z, N = S.N;
        const int ldr = N / 2 + 128, ldp = N + 128;  // row image: N/2 bytes per row; pair image: N bytes per pair-row
        std::vector<unsigned char*> img_r(NIMG), img_p(NIMG);
        for (int i = 0; i < NIMG; ++i) {
            CK(hipMalloc(&img_r[i], (size_t)Z * ldr)); CK(hipMalloc(&img_p[i], (size_t)(Z / 2) * ldp));
            hipLaunchKernelGGL(fill_bytes, dim3(2048), dim3(256), 0, st, img_r[i], (size_t)Z * ldr, 7u + i);
            hipLaunchKernelGGL(fill_bytes, dim3(2048), dim3(256), 0, st, img_p[i], (size_t)(Z / 2) * ldp, 77u + i);
        }
        uint16_t* x; CK(hipMalloc(&x, Z * 2));
        hipLaunchKernelGGL(fill_x, dim3((Z + 255) / 256), dim3(256), 0, st, x, Z, 3u);
        uint16_t* sz; CK(hipMalloc(&sz, (size_t)(Z / 32) * N * 4));
        hipLaunchKernelGGL(fill_sz, dim3(2048), dim3(256), 0, st, sz, (size_t)(Z / 32) * N * 2, 5u);
        float* slabs; CK(hipMalloc(&slabs, (size_t)8 * N * 4)); CK(hipMemsetAsync(slabs, 0, (size_t)8 * N * 4, st));
        double* yref; CK(hipMalloc(&yref, (size_t)N * 8));
        const size_t wsb = teal_workspace_bytes(Z, N);
        void* wsp; CK(hipMalloc(&wsp, wsb)); CK(hipMemsetAsync(wsp, 0, wsb, st));
        TK(teal_workspace_init(wsp, wsb, st));
        const int tiles = N / 256, nunits = Z / 32;
        auto time_graph = [&](auto&& launch) {
            hipGraph_t g; hipGraphExec_t ge;
            launch(0); CK(hipStreamSynchronize(st));
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < 32; ++i) launch(i % NIMG);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1000.0 / (20 * 32);
        };
        printf("-- %s\n", S.name);
        // product: row-pair image, 128-column tiles
        int nsl = 0;
        auto prod = [&](int i) {
            teal_gemv_in_t in; memset(&in, 0, sizeof in); in.mode = TEAL_IN_PLAIN; in.x = x;
            teal_gemv_out_t o; memset(&o, 0, sizeof o);
            o.nseg = 1; o.w[0] = img_p[i]; o.ld[0] = ldp; o.col0[0] = 0; o.ncols[0] = N; o.tau[0] = tau; o.mode = TEAL_OUT_SLABS;
            o.slabs = slabs; o.slabs_bytes = (size_t)8 * N * 4; o.slabs_interleaved = 1; o.weight_bits = 4; o.scale[0] = sz; o.scale_ld[0] = N; o.groupsize = 32;
            TK(teal_fused_gemv(&in, &o, Z, TEAL_F16, wsp, wsb, &nsl, st));
        };
        const double t_prod = time_graph(prod);
        printf("   product  (row pairs, 128-column tiles): %7.2f us   [%s]\n", t_prod, teal_last_launch_desc());
        // probe: row-granular image, 256-column tiles, a few row-slice counts
        for (int split = 1; split <= 8; ++split) {
            if (tiles * split > 2 * ncu || split * 16 > nunits) continue;
            if (tiles * split * 2 < ncu && split < 8 && tiles * (split + 1) <= 2 * ncu && (split + 1) * 16 <= nunits) continue;  // too few workgroups
            const int stride = (split + 3) & ~3;
            auto probe = [&](int i) {
                hipLaunchKernelGGL(i4rg_kernel, dim3(tiles, split), dim3(1024), 0, st, x, img_r[i], sz, slabs, Z, N, ldr, tau, stride);
            };
            // correctness on image 0
            probe(0);
            hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256), dim3(256), 0, st, x, img_r[0], sz, yref, Z, N, ldr, tau);
            CK(hipStreamSynchronize(st));
            std::vector<float> hs((size_t)N * stride); std::vector<double> hr(N);
            CK(hipMemcpy(hs.data(), slabs, hs.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), yref, (size_t)N * 8, hipMemcpyDeviceToHost));
            double worst = 0, scale = 0;
            for (int n = 0; n < N; ++n) { double s = 0; for (int k = 0; k < split; ++k) s += hs[(size_t)n * stride + k]; worst = std::max(worst, std::fabs(s - hr[n])); scale = std::max(scale, std::fabs(hr[n])); }
            const double t = time_graph(probe);
            printf("   probe    (rows, 256-column tiles) %3d tiles x %d slices = %3d workgroups: %7.2f us   (%+.1f %% vs product; max |err| %.2e of %.1f)\n", tiles, split,
                   tiles * split, t, (t / t_prod - 1.0) * 100.0, worst, scale);
        }
        for (int i = 0; i < NIMG; ++i) { CK(hipFree(img_r[i])); CK(hipFree(img_p[i])); }
        CK(hipFree(x)); CK(hipFree(sz)); CK(hipFree(slabs)); CK(hipFree(yref));
        TK(teal_workspace_release(wsp)); CK(hipFree(wsp));
    }
    return 0;
}
