// One-way cross-workgroup signal latency on MI355X: producers write a tile, release-fence, bump a per-unit counter
// in device memory; consumers (already resident, later block indices) spin on the counter, acquire, read the tile.
// Reports (consumer sees the data) - (producer finished its compute), in 10 ns ticks of wall_clock64().
// Build: hipcc --offload-arch=gfx950 -O3 signal_latency.hip -o signal_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(1024) void k(int nprod, int per_unit, unsigned* cnt, float* data, unsigned long long* t_prod,
                                           unsigned long long* t_cons, float* sink, int spin_iters) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b < nprod) {
        // fake work: ~spin_iters cycles
        float a = tid;
        for (int i = 0; i < spin_iters; ++i) a = a * 1.0001f + 0.5f;
        if (a == 12345.678f) sink[0] = a;
        if (tid < 64) data[b * 64 + tid] = a;
        __syncthreads();
        if (tid == 0) {
            t_prod[b] = wall_clock64();
            __threadfence();
            atomicAdd(&cnt[b / per_unit], 1u);
        }
    } else {
        const int u = b - nprod;  // consumer of unit u
        if (tid == 0) {
            int guard = 0;
            while (__hip_atomic_load(&cnt[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)per_unit && ++guard < (1 << 22)) {}
            __threadfence();
        }
        __syncthreads();
        float s = 0.f;
        if (tid < 64) for (int j = 0; j < per_unit; ++j) s += data[(u * per_unit + j) * 64 + tid];
        if (s == 12345.678f) sink[1] = s;
        __syncthreads();
        if (tid == 0) { t_cons[u] = wall_clock64(); cnt[u] = 0; }
    }
}

int main() {
    const int nprod = 192, per_unit = 6, ncons = nprod / per_unit;
    unsigned* cnt; float *data, *sink; unsigned long long *tp, *tc;
    hipMalloc(&cnt, ncons * 4); hipMemset(cnt, 0, ncons * 4);
    hipMalloc(&data, nprod * 64 * 4); hipMalloc(&sink, 8);
    hipMalloc(&tp, nprod * 8); hipMalloc(&tc, ncons * 8);
    std::vector<unsigned long long> hp(nprod), hc(ncons);
    for (int it = 0; it < 6; ++it) {
        hipLaunchKernelGGL(k, dim3(nprod + ncons), dim3(1024), 0, 0, nprod, per_unit, cnt, data, tp, tc, sink, 3000);
        hipDeviceSynchronize();
        hipMemcpy(hp.data(), tp, nprod * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), tc, ncons * 8, hipMemcpyDeviceToHost);
        std::vector<double> lat;
        for (int u = 0; u < ncons; ++u) {
            unsigned long long last = 0;
            for (int j = 0; j < per_unit; ++j) last = std::max(last, hp[u * per_unit + j]);
            lat.push_back(((double)hc[u] - (double)last) * 10.0 / 1e3);
        }
        std::sort(lat.begin(), lat.end());
        printf("iter %d: signal->consumed latency us: min %.2f median %.2f max %.2f\n", it, lat.front(), lat[lat.size() / 2], lat.back());
    }
    return 0;
}
