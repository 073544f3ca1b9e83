// teal_gemv_w16_bf16.hip — sparse_gemv_kernel instantiations: 16-bit weights, bf16 activations.
#include "teal_gemv_kernel.h"

namespace teal {
hipError_t launch_gemv_w16_bf16(const Params& p, size_t lds, const Config& c, hipStream_t st) {
    return launch_gemv_q<true, false>(p, lds, c, st);
}
}  // namespace teal
