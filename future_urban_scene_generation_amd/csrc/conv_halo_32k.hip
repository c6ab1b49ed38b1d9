// BN = 32 column tile with K split over the four waves (conv_kernel_halo.h, KS): k x k layers with few output channels.
#include "conv_kernel_halo.h"
namespace fusg {
hipError_t launch_halo_32k(const HaloK& k, dim3 grid, hipStream_t s, int pk, int mode) { return launch_halo<4,1,1,1,4>(k, grid, s, pk, mode); }
}  // namespace fusg
