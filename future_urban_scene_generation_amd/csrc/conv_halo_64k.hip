// BN = 64 column tile with K split over two waves per 32-column tile (conv_kernel_halo.h, KS).
#include "conv_kernel_halo.h"
namespace fusg {
hipError_t launch_halo_64k(const HaloK& k, dim3 grid, hipStream_t s, int pk, int mode) { return launch_halo<4,1,1,2,2>(k, grid, s, pk, mode); }
}  // namespace fusg
