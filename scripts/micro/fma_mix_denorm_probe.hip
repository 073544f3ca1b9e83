// Does v_fma_mix_f32 treat fp16 subnormal SOURCES like v_cvt_f32_f16 + v_fma_f32 does?  Every fp16 bit pattern a, a handful of
// b (fp16) and c (fp32): r_mix = v_fma_mix_f32(a.h, b.h, c) against r_cvt = v_fma_f32(cvt(a), cvt(b), c), compared bit for bit;
// the same for an fp32 subnormal c and results in the fp32 subnormal range.  Build: hipcc --offload-arch=gfx950 -O2 -o
// scripts/micro/fma_mix_denorm_probe scripts/micro/fma_mix_denorm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void probe(const uint16_t* b_bits, const float* c_vals, int nb, int nc, uint32_t* mix, uint32_t* cvt) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 65535
    for (int ib = 0; ib < nb; ++ib)
        for (int ic = 0; ic < nc; ++ic) {
            const uint32_t av = a, bv = b_bits[ib];
            const float c = c_vals[ic];
            float r1, r2, fa, fb;
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r1) : "v"(av), "v"(bv), "v"(c));
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(fa) : "v"(av));
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(fb) : "v"(bv));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r2) : "v"(fa), "v"(fb), "v"(c));
            const size_t o = ((size_t)ib * nc + ic) * 65536 + a;
            mix[o] = __builtin_bit_cast(uint32_t, r1);
            cvt[o] = __builtin_bit_cast(uint32_t, r2);
        }
}

static uint16_t h(float f) { _Float16 x = (_Float16)f; uint16_t u; memcpy(&u, &x, 2); return u; }

int main() {
    std::vector<uint16_t> b = {h(1.0f), h(0.5f), h(3.0f), h(-1.5f), h(1024.0f), 0x0001, 0x03ff, 0x0200, h(6.1e-5f), h(0.02f)};
    std::vector<float> c = {0.0f, 1.0f, -0.25f, 1e-30f, 1e-40f /* fp32 subnormal */, 3.0e-6f};
    const int nb = b.size(), nc = c.size();
    uint16_t* db; float* dc; uint32_t *dm, *dv;
    hipMalloc(&db, nb * 2); hipMalloc(&dc, nc * 4);
    hipMalloc(&dm, (size_t)nb * nc * 65536 * 4); hipMalloc(&dv, (size_t)nb * nc * 65536 * 4);
    hipMemcpy(db, b.data(), nb * 2, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), nc * 4, hipMemcpyHostToDevice);
    probe<<<256, 256>>>(db, dc, nb, nc, dm, dv);
    std::vector<uint32_t> m((size_t)nb * nc * 65536), v(m.size());
    hipMemcpy(m.data(), dm, m.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(v.data(), dv, v.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0, diff_sub_a = 0, diff_sub_b = 0, diff_other = 0, shown = 0;
    for (int ib = 0; ib < nb; ++ib)
        for (int ic = 0; ic < nc; ++ic)
            for (uint32_t a = 0; a < 65536; ++a) {
                const size_t o = ((size_t)ib * nc + ic) * 65536 + a;
                if (m[o] == v[o]) continue;
                if (((a >> 10) & 31) == 31) continue;  // inf / nan payloads
                ++diff;
                const bool sa = ((a >> 10) & 31) == 0 && (a & 1023), sb = ((b[ib] >> 10) & 31) == 0 && (b[ib] & 1023);
                if (sa) ++diff_sub_a; else if (sb) ++diff_sub_b; else ++diff_other;
                if (shown < 12) { printf("  a 0x%04x b 0x%04x c %g: mix 0x%08x cvt+fma 0x%08x\n", a, b[ib], c[ic], m[o], v[o]); ++shown; }
            }
    printf("v_fma_mix_f32 vs v_cvt_f32_f16 + v_fma_f32 over %zu finite cases: %zu differ (a subnormal: %zu, b subnormal only: %zu, neither: %zu)\n",
           m.size(), diff, diff_sub_a, diff_sub_b, diff_other);
    return 0;
}
