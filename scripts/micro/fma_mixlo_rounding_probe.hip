// Is v_fma_mixlo_f16 (fp32 fma, result written as fp16) the same as v_fma_f32 followed by v_cvt_f16_f32?  The compiler folds
// `(_Float16)fmaf(a, b, c)` into the former where it can (after -packed-fp32-ops it did so in decode_attention_split_kernel's
// RoPE but not in the GEMV epilogue's).  Random fp16 a, b and fp32 c (c = -(x1 * s) as in rope_even), 2^26 cases.
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/micro/fma_mixlo_rounding_probe scripts/micro/fma_mixlo_rounding_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

__global__ void probe(unsigned long long* counts, uint32_t* examples) {
    uint32_t s = 0x9E3779B9u * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long differ = 0, tie_cases = 0;
    for (int it = 0; it < 1024; ++it) {
        // fp16 values of magnitude 2^-4 .. 2^3 (exponent field 11..18), random mantissas and signs
        auto h16 = [&]() { const uint32_t r = rng(s); return (r & 0x83FFu) | ((11u + (r >> 16) % 8u) << 10); };
        const uint32_t a = h16(), b = h16(), x1 = h16(), sn = h16();
        float fx1, fsn, c, r_fma, lo_cvt;
        uint32_t lo_mix;
        asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(fx1) : "v"(x1));
        asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(fsn) : "v"(sn));
        c = -(fx1 * fsn);
        asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(lo_mix) : "v"(a), "v"(b), "v"(c), "0"(0u));
        asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r_fma) : "v"(a), "v"(b), "v"(c));
        uint32_t lo2;
        asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(lo2) : "v"(r_fma));
        (void)lo_cvt;
        // fp32 results that sit exactly on an fp16 rounding tie (13 dropped bits = 1 0000 0000 0000): where a single rounding
        // of the exact result could go the other way
        if ((__builtin_bit_cast(uint32_t, r_fma) & 0x1FFFu) == 0x1000u) ++tie_cases;
        if ((lo_mix & 0xFFFFu) != (lo2 & 0xFFFFu)) {
            if (differ == 0) { examples[4 * (blockIdx.x * blockDim.x + threadIdx.x) + 0] = a | (b << 16); examples[4 * (blockIdx.x * blockDim.x + threadIdx.x) + 1] = __builtin_bit_cast(uint32_t, c);
                               examples[4 * (blockIdx.x * blockDim.x + threadIdx.x) + 2] = lo_mix & 0xFFFFu; examples[4 * (blockIdx.x * blockDim.x + threadIdx.x) + 3] = lo2 & 0xFFFFu; }
            ++differ;
        }
    }
    atomicAdd(&counts[0], differ);
    atomicAdd(&counts[1], tie_cases);
}

int main() {
    unsigned long long* dc; uint32_t* de;
    const int nb = 256, nt = 256;
    (void)hipMalloc(&dc, 16); (void)hipMemset(dc, 0, 16);
    (void)hipMalloc(&de, nb * nt * 16); (void)hipMemset(de, 0, nb * nt * 16);
    probe<<<nb, nt>>>(dc, de);
    unsigned long long c[2];
    (void)hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    static uint32_t ex[256 * 256 * 4];
    (void)hipMemcpy(ex, de, sizeof(ex), hipMemcpyDeviceToHost);
    printf("v_fma_mixlo_f16 vs v_fma_mix_f32 + v_cvt_f16_f32 over %d cases: %llu differ; fp32 results exactly on an fp16 tie: %llu\n", nb * nt * 1024, c[0], c[1]);
    int shown = 0;
    for (int i = 0; i < nb * nt && shown < 6; ++i)
        if (ex[4 * i + 2] != ex[4 * i + 3]) { printf("  a 0x%04x b 0x%04x c 0x%08x: mixlo 0x%04x fma+cvt 0x%04x\n", ex[4 * i] & 0xFFFF, ex[4 * i] >> 16, ex[4 * i + 1], ex[4 * i + 2], ex[4 * i + 3]); ++shown; }
    return 0;
}
