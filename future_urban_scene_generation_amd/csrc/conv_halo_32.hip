// BN = 32 column tile of the halo-tiled split-fp16 / bf16 convolution (see conv_kernel_halo.h).
#include "conv_kernel_halo.h"
namespace fusg {
hipError_t launch_halo_32(const HaloK& k, dim3 grid, hipStream_t s, int pk, int mode) { return launch_halo<1,1,4,1>(k, grid, s, pk, mode); }
}  // namespace fusg
