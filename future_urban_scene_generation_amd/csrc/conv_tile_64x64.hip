// 64x64 workgroup tile of the fused implicit-GEMM convolution (see conv_kernel.h).
#include "conv_kernel.h"
namespace fusg {
hipError_t launch_tile_64x64(const ConvK& k, dim3 grid, hipStream_t s, int pk, bool gen) {
    return launch_tile<1, 1, 2, 2>(k, grid, s, pk, gen);
}
}  // namespace fusg
