// Exact-fp32 tap-unit convolution for the few-channel stems (conv_kernel_tapunit_f32.h): the three column-tile widths.
#include "conv_kernel_tapunit_f32.h"
namespace fusg {
hipError_t launch_tapunit_f32_128(const TapUnitF& k, dim3 grid, hipStream_t s, int pk) { return launch_tapunit_f32<2, 2, 2, 2>(k, grid, s, pk); }
hipError_t launch_tapunit_f32_64(const TapUnitF& k, dim3 grid, hipStream_t s, int pk) { return launch_tapunit_f32<2, 1, 2, 2>(k, grid, s, pk); }
hipError_t launch_tapunit_f32_32(const TapUnitF& k, dim3 grid, hipStream_t s, int pk) { return launch_tapunit_f32<1, 1, 4, 1>(k, grid, s, pk); }
}  // namespace fusg
