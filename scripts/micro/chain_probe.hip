// chain_probe — what does an EARLY-RESIDENT consumer buy over a plain kernel boundary on MI355X?  (round-3 verdict, lever b)
//
// A chain of dependent "GEMV-like" steps: every workgroup reads the previous step's whole output vector (all-to-all, like
// the RESID_NORM producer), streams its share of a weight image non-temporally, reduces, publishes 64 floats.  Timed as
//   A  one stream, plain launches (the boundary is the hand-off: what the decode step does today)
//   B  two streams, steps alternating between them with NO graph edge between consecutive steps: step k+1 is dispatched
//      while step k runs (two 16-wave workgroups fit a CU: <= 64 VGPRs), sets itself up and polls per-XCD completion
//      counters that step k's workgroups bump after draining their write-through output stores
//   C  protocol B on ONE stream (the pure cost of the signalling: no overlap possible)
// Benchmark utility, not product code.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/chain_probe.hip -o scripts/micro/chain_probe
//   chain_probe [steps 40] [MB per step 50] [wgs 256] [reps 200]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// mode 0: plain (the kernel boundary orders everything); mode 1: wait for cnt_in, signal cnt_out
__global__ __launch_bounds__(1024, 8) void step_kernel(const u32x4* __restrict__ w, const int loads_per_lane, const float* in, float* out,
                                                       unsigned* cnt_in, unsigned* cnt_out, const int per_xcd, const int mode,
                                                       unsigned long long* stamps, unsigned* err) {
    __shared__ float red[16 * 64];
    __shared__ int okflag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, bid = blockIdx.x;
    const unsigned long long t0 = wall_clock64();
    if (mode == 1 && cnt_in) {
        if (wave == 0) {
            bool ok = false;
            const unsigned long long tlim = t0 + 200000ull;  // 2 ms at 100 MHz: give up (never hang the box)
            while (true) {
                unsigned v = per_xcd;
                if (lane < 8) v = __hip_atomic_load(cnt_in + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(v >= (unsigned)per_xcd)) { ok = true; break; }
                if (wall_clock64() > tlim) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok && lane == 0) atomicAdd(err, 1u);
        }
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    // all-to-all read of the previous step's vector (4096 floats = 16 KB), L1-bypassing like the slab reads
    float xv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = __hip_atomic_load(in + k * 1024 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float s = (xv[0] + xv[1]) + (xv[2] + xv[3]);
    // a dependent "list": the stream's addresses depend on the activation (never actually moves)
    const size_t shift = (s == 1234.5678f) ? 1 : 0;
    const u32x4* p = w + (size_t)bid * 1024 * (size_t)loads_per_lane + tid + shift;
    const unsigned long long t2 = wall_clock64();
    u32x4 acc = {0, 0, 0, 0};
    int i = 0;
    for (; i + 4 <= loads_per_lane; i += 4) {
        u32x4 a = __builtin_nontemporal_load(p + (size_t)(i + 0) * 1024);
        u32x4 b = __builtin_nontemporal_load(p + (size_t)(i + 1) * 1024);
        u32x4 c = __builtin_nontemporal_load(p + (size_t)(i + 2) * 1024);
        u32x4 d = __builtin_nontemporal_load(p + (size_t)(i + 3) * 1024);
        acc ^= a; acc ^= b; acc ^= c; acc ^= d;
    }
    float part = __uint_as_float((acc.x ^ acc.y ^ acc.z ^ acc.w) & 0x007fffffu) * 1e-30f + s * 1e-3f;
    red[wave * 64 + lane] = part;
    __syncthreads();
    const unsigned long long t3 = wall_clock64();
    if (tid < 64) {
        float sum = 0.0f;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) sum += red[wv * 64 + tid];
        // every workgroup owns 16 of the 4096 output floats (256 workgroups); lanes 16.. idle
        if (tid < 16) {
            if (mode == 1) __hip_atomic_store(out + bid * 16 + tid, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else out[bid * 16 + tid] = sum;
        }
        if (mode == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) __hip_atomic_fetch_add(cnt_out + (bid & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (stamps && tid == 0) {
        unsigned long long* r = stamps + (size_t)bid * 8;
        r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3; r[4] = wall_clock64();
    }
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 40;
    const int mb = argc > 2 ? atoi(argv[2]) : 50;
    const int wgs = argc > 3 ? atoi(argv[3]) : 256;
    const int reps = argc > 4 ? atoi(argv[4]) : 200;
    const int loads_per_lane = (int)(((size_t)mb * 1000000 / wgs / 1024 / 16) & ~3ull);
    const size_t wbytes = (size_t)wgs * 1024 * loads_per_lane * 16 + 64;
    const int nw = 24;  // rotate over weight images so that nothing is cache resident (24 x 50 MB > 256 MB MALL)
    std::vector<u32x4*> W(nw);
    for (auto& p : W) { CK(hipMalloc(&p, wbytes)); CK(hipMemset(p, 1, wbytes)); }
    float* buf[2]; CK(hipMalloc(&buf[0], 4096 * 4)); CK(hipMalloc(&buf[1], 4096 * 4));
    CK(hipMemset(buf[0], 0, 4096 * 4)); CK(hipMemset(buf[1], 0, 4096 * 4));
    unsigned* cnt; CK(hipMalloc(&cnt, (size_t)(steps + 1) * 8 * 4));
    unsigned* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)steps * wgs * 8 * 8)); CK(hipMemset(stamps, 0, (size_t)steps * wgs * 8 * 8));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, step_kernel, 1024, 0));
    printf("chain_probe: %d steps x %d MB over %d workgroups (%d x 16-byte loads per lane), occupancy %d workgroups / CU\n", steps, mb, wgs,
           loads_per_lane, occ);
    const int per_xcd = wgs / 8;

    auto build = [&](int variant, bool with_stamps) {
        hipGraph_t g; hipGraphExec_t ge;
        hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
        if (variant != 0) CK(hipMemsetAsync(cnt, 0, (size_t)(steps + 1) * 8 * 4, s0));
        if (variant == 1) { CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0)); }
        for (int k = 0; k < steps; ++k) {
            hipStream_t st = (variant == 1 && (k & 1)) ? s1 : s0;
            const int mode = variant == 0 ? 0 : 1;
            hipLaunchKernelGGL(step_kernel, dim3(wgs), dim3(1024), 0, st, W[k % nw], loads_per_lane, buf[(k + 1) & 1], buf[k & 1],
                               (mode && k) ? cnt + (size_t)k * 8 : nullptr, cnt + (size_t)(k + 1) * 8, per_xcd, mode,
                               with_stamps ? stamps + (size_t)k * wgs * 8 : nullptr, err);
        }
        if (variant == 1) { CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); }
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        return ge;
    };
    auto time_graph = [&](hipGraphExec_t ge, int n) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(e0, s0));
        for (int i = 0; i < n; ++i) CK(hipGraphLaunch(ge, s0));
        CK(hipEventRecord(e1, s0));
        CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1000.0 / n / steps;
    };
    const char* names[3] = {"A one stream, plain boundaries", "B two streams, completion counters", "C one stream, completion counters"};
    hipGraphExec_t ge[3];
    if (occ < 2) { printf("occupancy < 2: variant B could deadlock, not run\n"); return 1; }
    for (int v = 0; v < 3; ++v) ge[v] = build(v, false);
    for (int r = 0; r < 4; ++r) {
        printf("round %d:", r);
        for (int v = 0; v < 3; ++v) printf("  %c %.2f us/step", 'A' + v, time_graph(ge[v], reps));
        printf("\n");
    }
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("poll timeouts: %u\n", herr);
    // stamps: when do the consumers enter, get released, and end, relative to the previous step's last end?
    std::vector<unsigned long long> hs((size_t)steps * wgs * 8);
    for (int v = 0; v < 3; ++v) {
        hipGraphExec_t gs = build(v, true);
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(gs, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        double d_entry = 0, d_rel = 0, d_list = 0, d_str = 0, d_end = 0, d_step = 0; int n = 0;
        for (int k = 2; k < steps; ++k) {
            auto col = [&](int kk, int c, bool mx) { unsigned long long m = mx ? 0 : ~0ull; for (int b = 0; b < wgs; ++b) { unsigned long long x = hs[((size_t)kk * wgs + b) * 8 + c]; m = mx ? std::max(m, x) : std::min(m, x); } return (double)m; };
            const double prev_end = col(k - 1, 4, true);
            d_entry += (col(k, 0, true) - prev_end);   // last workgroup entry vs previous step's last end (negative = early resident)
            d_rel += (col(k, 1, true) - prev_end);     // last workgroup released (poll done) vs previous end
            d_list += (col(k, 2, true) - prev_end);    // activation loaded
            d_str += (col(k, 3, true) - col(k, 2, true));
            d_end += (col(k, 4, true) - col(k, 3, true));
            d_step += (col(k, 4, true) - prev_end);
            ++n;
        }
        if (getenv("CP_TIMELINE")) {
            double base = 1e30; for (int b = 0; b < wgs; ++b) base = std::min(base, (double)hs[(size_t)b * 8]);
            for (int k = 0; k < std::min(steps, 10); ++k) {
                auto col = [&](int c, bool mx) { double m = mx ? 0 : 1e30; for (int b = 0; b < wgs; ++b) { double x = (double)hs[((size_t)k * wgs + b) * 8 + c]; m = mx ? std::max(m, x) : std::min(m, x); } return (m - base) / 100.0; };
                printf("   step %d: entry %.2f..%.2f  released %.2f..%.2f  stream start ..%.2f  stream end %.2f..%.2f  end ..%.2f\n", k, col(0, false), col(0, true), col(1, false), col(1, true), col(2, true), col(3, false), col(3, true), col(4, true));
            }
        }
        printf("%s\n   vs previous step's last end (us): last entry %+.2f, last released %+.2f, activation in %+.2f | stream %.2f, tail %.2f | step %.2f\n",
               names[v], d_entry / n / 100, d_rel / n / 100, d_list / n / 100, d_str / n / 100, d_end / n / 100, d_step / n / 100);
    }
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("poll timeouts (total): %u\n", herr);
    // D: eager launches alternating between two streams (no graph), counters; E: eager, one stream, plain
    for (int v = 0; v < 2; ++v) {
        auto run = [&](bool with_stamps) {
            if (v == 0) CK(hipMemsetAsync(cnt, 0, (size_t)(steps + 1) * 8 * 4, s0));
            hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
            if (v == 0) { CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0)); }
            for (int k = 0; k < steps; ++k) {
                hipStream_t st = (v == 0 && (k & 1)) ? s1 : s0;
                const int mode = v == 0 ? 1 : 0;
                hipLaunchKernelGGL(step_kernel, dim3(wgs), dim3(1024), 0, st, W[k % nw], loads_per_lane, buf[(k + 1) & 1], buf[k & 1],
                                   (mode && k) ? cnt + (size_t)k * 8 : nullptr, cnt + (size_t)(k + 1) * 8, per_xcd, mode,
                                   with_stamps ? stamps + (size_t)k * wgs * 8 : nullptr, err);
            }
            if (v == 0) { CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); }
            CK(hipEventDestroy(ef)); CK(hipEventDestroy(ej));
        };
        for (int i = 0; i < 3; ++i) run(false);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s0));
        const int n = 20;
        for (int i = 0; i < n; ++i) run(false);
        CK(hipEventRecord(e1, s0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.2f us/step\n", v == 0 ? "D eager, two streams, completion counters" : "E eager, one stream, plain", ms * 1000.0 / n / steps);
        run(true);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        double base = 1e30; for (int b = 0; b < wgs; ++b) base = std::min(base, (double)hs[(size_t)b * 8]);
        for (int k = 0; k < std::min(steps, 10); ++k) {
            auto col = [&](int c, bool mx) { double m = mx ? 0 : 1e30; for (int b = 0; b < wgs; ++b) { double x = (double)hs[((size_t)k * wgs + b) * 8 + c]; m = mx ? std::max(m, x) : std::min(m, x); } return (m - base) / 100.0; };
            printf("   step %d: entry %.2f..%.2f  released %.2f..%.2f  stream start ..%.2f  stream end %.2f..%.2f  end ..%.2f\n", k, col(0, false), col(0, true), col(1, false), col(1, true), col(2, true), col(3, false), col(3, true), col(4, true));
        }
    }
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("poll timeouts (total): %u\n", herr);
    return 0;
}
