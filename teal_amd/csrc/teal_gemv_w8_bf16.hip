// teal_gemv_w8_bf16.hip — sparse_gemv_kernel instantiations: int8 weight-only weights, bf16 activations.
#include "teal_gemv_kernel.h"

namespace teal {
hipError_t launch_gemv_w8_bf16(const Params& p, size_t lds, const Config& c, hipStream_t st) {
    return launch_gemv_q<true, true>(p, lds, c, st);
}
}  // namespace teal
