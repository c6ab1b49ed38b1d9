// BN = 128 column tile of the halo-tiled split-fp16 / bf16 convolution (see conv_kernel_halo.h).
// Wave layout 1 x 4: every wave computes all 128 pixels of the patch for 32 of the 128 columns, so no weight fragment is
// fetched by two waves (the 2 x 2 layout of rounds 1-2 fetched each twice: 8 KiB per wave and (chunk, tap) step from
// L2 / L1, against 16 KiB of LDS reads here).  Same MFMAs, same registers; A/B on one card, whole pass (round 3): conv time
// 24.28 -> 23.54 ms, 1383 -> 1424 crops/s; the bf16 mode does not care (18.16 vs 18.19 ms).  -DFUSG_HALO_W22 builds 2 x 2.
#include "conv_kernel_halo.h"
namespace fusg {
hipError_t launch_halo_128(const HaloK& k, dim3 grid, hipStream_t s, int pk, int mode) {
#ifdef FUSG_HALO_W22
    return launch_halo<2,2,2,2>(k, grid, s, pk, mode);
#else
    return launch_halo<4,1,1,4>(k, grid, s, pk, mode);
#endif
}
}  // namespace fusg
#ifdef FUSG_HALO_STAMPS
extern "C" int fusg_debug_halo_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fusg::g_halo_stamps), sizeof(unsigned long long) * 64 * 4 * 40) == hipSuccess ? 0 : 1;
}
extern "C" int fusg_debug_halo_stamps_clear(void) {
    static unsigned long long z[64 * 4 * 40];
    return hipMemcpyToSymbol(HIP_SYMBOL(fusg::g_halo_stamps), z, sizeof z) == hipSuccess ? 0 : 1;
}
#endif
