// teal_gemv_w16_f16.hip — sparse_gemv_kernel instantiations: 16-bit weights, fp16 activations.
#include "teal_gemv_kernel.h"

namespace teal {
hipError_t launch_gemv_w16_f16(const Params& p, size_t lds, const Config& c, hipStream_t st) {
    return launch_gemv_q<false, false>(p, lds, c, st);
}
}  // namespace teal
