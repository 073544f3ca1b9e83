// teal_gemv_fast_w8_bf16.hip — gemv_fast_kernel instantiations: int8 weight-only images, bf16 activations.
// Compiled with -mllvm -amdgpu-kernarg-preload-count=11 (teal_amd/_lib.py): the scalar kernel parameters are preloaded.
#include "teal_gemv_fast.h"

namespace teal {
hipError_t launch_fast_w8_bf16(const FastLaunch& f, hipStream_t st) {
    return launch_fast_w8_q<true>(f, st);
}
}  // namespace teal
