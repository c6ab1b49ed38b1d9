// BN = 128 column tile of the tap-unit convolution for few-channel stems (see conv_kernel_tapunit.h).
#include "conv_kernel_tapunit.h"
namespace fusg {
hipError_t launch_tapunit_128(const TapUnitK& k, dim3 grid, hipStream_t s, int pk, int unit, int mode) { return launch_tapunit<2, 2, 2, 2>(k, grid, s, pk, unit, mode); }
}  // namespace fusg
