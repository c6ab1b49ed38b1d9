// 64x128 workgroup tile of the split-fp16 (f16x3) implicit-GEMM convolution (see conv_kernel_h3.h).
#include "conv_kernel_h3.h"
namespace fusg {
hipError_t launch_h3_64x128(const ConvK& k, dim3 grid, hipStream_t s, int pk, bool gen) {
    return launch_h3<1, 2, 2, 2>(k, grid, s, pk, gen);
}
}  // namespace fusg
