// cu_cap_probe — what bounds the bytes a compute unit pulls from HBM on MI355X?
// Round 2 measured ~38 GB/s per CU for the sparse GEMV's stream loop whatever the row-segment size, so a launch with
// 172 workgroups (gate|up of Llama-2-7B) streams at ~6.5 TB/s while 256 workgroups reach ~7.2.  This probe separates the
// candidates: a cap per CU (then only more CUs help), per workgroup / wave slot (then two co-resident workgroups per CU
// help), or bytes in flight x latency (then deeper pipelines help).
//   contiguous reads, 16 B per lane, U loads in flight per lane, W waves per workgroup, G workgroups;
//   "x2" rows: workgroups of 8 waves, so that two of them fit a CU's 16 wave slots... (VGPR use is tiny: up to 8 waves
//   per SIMD fit; residency is then bounded by the grid, G <= 2 x 256).
// hipcc --offload-arch=gfx950 -O3 scripts/micro/cu_cap_probe.hip -o scripts/micro/cu_cap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(NT) void contig(const char* __restrict__ w, size_t bytes_per_wg, unsigned* sink) {
    const char* p = w + (size_t)blockIdx.x * bytes_per_wg + threadIdx.x * 16;
    u32x4 acc = {0, 0, 0, 0};
    constexpr size_t STEP = (size_t)NT * 16;
    for (size_t o = 0; o + U * STEP <= bytes_per_wg; o += U * STEP) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + o + u * STEP));
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

// the GEMV's gather: workgroup t reads SEG bytes of every row (row stride ldb), 1024 B per wave instruction,
// i.e. 1024 / SEG rows; rows dealt to the 16 waves; U loads in flight per lane, two batches (software pipeline)
template <int U, int SEG>
__global__ __launch_bounds__(1024) void gather(const char* __restrict__ w, size_t ldb, int rows, unsigned* sink) {
    constexpr int LPRL = SEG / 16, RPW = 64 / LPRL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane / LPRL, cl = lane % LPRL;
    const char* base = w + (size_t)blockIdx.x * SEG + cl * 16;
    u32x4 acc = {0, 0, 0, 0};
    const int per_wave = rows / 16;
    const int r0 = wave * per_wave;
    u32x4 a[U], b[U];
    auto issue = [&](u32x4 (&v)[U], int e) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (size_t)(r0 + e + u * RPW + g) * ldb));
    };
    auto eat = [&](u32x4 (&v)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    };
    constexpr int STEP = U * RPW;
    int e = 0;
    if (STEP <= per_wave) issue(a, 0);
    while (e + STEP <= per_wave) {
        const bool more = e + 2 * STEP <= per_wave;
        if (more) issue(b, e + STEP);
        eat(a);
        e += STEP;
        if (!more) break;
        const bool more2 = e + 2 * STEP <= per_wave;
        if (more2) issue(a, e + STEP);
        eat(b);
        e += STEP;
        if (!more2) break;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main() {
    const size_t BUF = (size_t)192 << 20;  // per buffer
    const int NBUF = 8;                    // 1.5 GB rotating: beyond the 256 MB Infinity Cache
    std::vector<char*> bufs(NBUF);
    for (auto& b : bufs) { CK(hipMalloc(&b, BUF)); CK(hipMemset(b, 1, BUF)); }
    unsigned* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& launch) {
        for (int i = 0; i < NBUF; ++i) launch(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 4 * NBUF;
        for (int i = 0; i < reps; ++i) launch(i % NBUF);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1000.0 / reps;
    };
    const double total = 90.0 * (1 << 20);  // ~ the gate|up launch
    printf("contiguous reads, ~90 MB per launch, us per launch incl. ~1.5 us launch boundary | TB/s | GB/s per workgroup\n");
    printf("%-28s", "workgroups:");
    const int Gs[] = {64, 128, 172, 192, 256, 344, 512, 768, 1024};
    for (int G : Gs) printf(" %14d", G);
    printf("\n");
#define ROW(U, NT)                                                                                                      \
    {                                                                                                                   \
        printf("U=%-2d loads/lane, %2d waves/wg ", U, NT / 64);                                                          \
        for (int G : Gs) {                                                                                              \
            const size_t step = (size_t)NT * 16 * U;                                                                    \
            const size_t per = (size_t)(total / G) / step * step;                                                       \
            const double us = timeit([&](int i) { hipLaunchKernelGGL((contig<U, NT>), dim3(G), dim3(NT), 0, 0, bufs[i], per, sink); }); \
            const double bytes = (double)per * G;                                                                       \
            printf(" %5.1f %4.2f %4.0f", us, bytes / us / 1e6, bytes / us / 1e3 / G);                                     \
        }                                                                                                               \
        printf("\n");                                                                                                   \
    }
    ROW(2, 1024) ROW(4, 1024) ROW(8, 1024) ROW(16, 1024)
    ROW(4, 512) ROW(8, 512) ROW(16, 512)
    ROW(8, 256) ROW(16, 256)
    printf("\ngather: 4096 rows x SEG bytes per workgroup, row stride 45184 B, us | TB/s | GB/s per workgroup\n");
    printf("%-28s", "workgroups:");
    const int Hs[] = {86, 172, 256, 344};
    for (int G : Hs) printf(" %14d", G);
    printf("\n");
#define GROW(U, SEG)                                                                                                    \
    {                                                                                                                   \
        printf("U=%-2d x 2 batches, SEG %-4d  ", U, SEG);                                                                \
        for (int G : Hs) {                                                                                              \
            const int rows = 4096;                                                                                      \
            if ((size_t)G * SEG > 45184) { printf(" %14s", "-"); continue; }                                              \
            const double us = timeit([&](int i) { hipLaunchKernelGGL((gather<U, SEG>), dim3(G), dim3(1024), 0, 0, bufs[i], (size_t)45184, rows, sink); }); \
            const double bytes = (double)rows * SEG * G;                                                                 \
            printf(" %5.1f %4.2f %4.0f", us, bytes / us / 1e6, bytes / us / 1e3 / G);                                     \
        }                                                                                                               \
        printf("\n");                                                                                                   \
    }
    GROW(4, 128) GROW(8, 128) GROW(4, 64) GROW(8, 64) GROW(4, 256) GROW(2, 256)
    return 0;
}
