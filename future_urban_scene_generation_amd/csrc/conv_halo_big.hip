// bf16 mode of the halo kernel on a 16 x 16 pixel patch x 128 columns (conv_kernel_halo.h, BM == 256; round 4, BASELINE configs[4]).
#include "conv_kernel_halo.h"
namespace fusg {
hipError_t launch_halo_big22(const HaloK& k, dim3 grid, hipStream_t s, int pk) { return launch_halo_big<4,2,2,2>(k, grid, s, pk); }
hipError_t launch_halo_big14(const HaloK& k, dim3 grid, hipStream_t s, int pk) { return launch_halo_big<8,1,1,4>(k, grid, s, pk); }
}  // namespace fusg
