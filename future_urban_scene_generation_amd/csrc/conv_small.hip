// Small-spatial split-fp16 convolution (conv_kernel_small.h): launcher.
#include "conv_kernel_small.h"
namespace fusg {
hipError_t launch_small(const SmallK& k, dim3 grid, hipStream_t s, int pk) {
    const size_t lds = small_lds_bytes(k.nchw, k.NPIX);
    const void* fn = pk == PK_NONE ? (const void*)conv_small_h3<PK_NONE> : pk == PK_ELU ? (const void*)conv_small_h3<PK_ELU>
                                                                                       : (const void*)conv_small_h3<PK_AFFINE>;
    if (hipError_t e = ensure_dyn_lds(fn, 144 * 1024); e != hipSuccess) return e;
    SmallK kk = k;
    kk.part_off = (int)((size_t)2 * k.nchw * k.NPIX * 32 * sizeof(_Float16));
    void* args[] = {(void*)&kk};
    return hipLaunchKernel(fn, grid, dim3(256), args, lds, s);
}
}  // namespace fusg
