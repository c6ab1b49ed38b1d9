// 128x32 workgroup tile of the split-fp16 (f16x3) implicit-GEMM convolution (see conv_kernel_h3.h).
#include "conv_kernel_h3.h"
namespace fusg {
hipError_t launch_h3_128x32(const ConvK& k, dim3 grid, hipStream_t s, int pk, bool gen) {
    return launch_h3<1, 1, 4, 1>(k, grid, s, pk, gen);
}
}  // namespace fusg
