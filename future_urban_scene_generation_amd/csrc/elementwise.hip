// HBM-bound elementwise / data-movement kernels of the hot path (gfx950).  NHWC activations are
// processed as float4 along channels (16 B per lane, coalesced); boundary tensors with arbitrary
// strides go through the generic per-pixel kernels.
#include "common.h"

namespace fusg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct T4 {                     // device view of a fusg_tensor
    char* p; long n, c, h, w, sn, sc, sh, sw;
};
static inline T4 view(const fusg_tensor& t) {
    T4 v; v.p = (char*)t.data; v.n = t.n; v.c = t.c; v.h = t.h; v.w = t.w; v.sn = t.sn; v.sc = t.sc; v.sh = t.sh; v.sw = t.sw;
    return v;
}
static inline unsigned blocks_for(long total, int per = 256) {
    long b = (total + per - 1) / per;
    return (unsigned)(b < 1 ? 1 : b);
}

__device__ __forceinline__ float act1(float v, int act) {
    switch (act) {
        case FUSG_ACT_RELU: return fmaxf(v, 0.f);
        case FUSG_ACT_TANH: return tanhf(v);
        case FUSG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case FUSG_ACT_TANH01: return (tanhf(v) + 1.f) / 2.f;
        default: return v;
    }
}

// dst = act(x*scale[b,c]+shift[b,c]) + res ; NHWC-physical, one thread per float4
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, int xCs, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, long bstride, int act,
                                                         const float* __restrict__ res, int rCs, float* __restrict__ dst,
                                                         int dCs, long HW, int C4, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % C4);
        const long pix = i / C4;
        const long b = pix / HW;
        f32x4 v = *(const f32x4*)(x + pix * xCs + q * 4);
        if (scale) {
            const f32x4 sc = *(const f32x4*)(scale + b * bstride + q * 4);
            const f32x4 sh = *(const f32x4*)(shift + b * bstride + q * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = fmaf(v[c], sc[c], sh[c]);
        }
        if (act != FUSG_ACT_NONE) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = act1(v[c], act);
        }
        if (res) v += *(const f32x4*)(res + pix * rCs + q * 4);
        *(f32x4*)(dst + pix * dCs + q * 4) = v;
    }
}

__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, int xCs, float* __restrict__ dst, int dCs,
                                                       int Ho, int Wo, int W, int C4, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % C4);
        long pix = i / C4;
        const int ox = (int)(pix % Wo); pix /= Wo;
        const int oy = (int)(pix % Ho);
        const long b = pix / Ho;
        const float* s = x + ((b * (2L * Ho) + 2 * oy) * W + 2 * ox) * xCs + q * 4;
        const f32x4 a = *(const f32x4*)s, bb = *(const f32x4*)(s + xCs);
        const f32x4 c = *(const f32x4*)(s + (long)W * xCs), d = *(const f32x4*)(s + (long)W * xCs + xCs);
        f32x4 m;
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(a[k], bb[k]), fmaxf(c[k], d[k]));
        *(f32x4*)(dst + ((b * Ho + oy) * (long)Wo + ox) * dCs + q * 4) = m;
    }
}

// dst[b,y,x,:] = up1[b,y,x,:] + low[b,y/2,x/2,:]
__global__ __launch_bounds__(256) void upsample2_add_kernel(const float* __restrict__ low, int lCs, const float* __restrict__ up1,
                                                            int uCs, float* __restrict__ dst, int dCs, int H, int W, int C4,
                                                            long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % C4);
        long pix = i / C4;
        const int x = (int)(pix % W); const long r = pix / W;
        const int y = (int)(r % H);
        const long b = r / H;
        const f32x4 l = *(const f32x4*)(low + ((b * (H >> 1) + (y >> 1)) * (long)(W >> 1) + (x >> 1)) * lCs + q * 4);
        const f32x4 u = *(const f32x4*)(up1 + pix * uCs + q * 4);
        *(f32x4*)(dst + pix * dCs + q * 4) = u + l;
    }
}

// generic strided kernels: one thread per pixel, loop over channels
__global__ __launch_bounds__(256) void copy4d_kernel(T4 s, T4 d, int cfill, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long x = i % s.w; const long r = i / s.w;
    const long y = r % s.h; const long b = r / s.h;
    const float* sp = (const float*)s.p + b * s.sn + y * s.sh + x * s.sw;
    float* dp = (float*)d.p + b * d.sn + y * d.sh + x * d.sw;
    for (long c = 0; c < s.c; ++c) dp[c * d.sc] = sp[c * s.sc];
    for (long c = s.c; c < cfill; ++c) dp[c * d.sc] = 0.f;
}

// one thread per element, channel fastest (coalesced for NHWC destinations; the Sampler's noise operand is a
// small NCHW tensor)
// NCHW-contiguous -> NHWC with zero-filled channel padding, through an LDS tile so that both the reads
// (256 consecutive pixels of one channel plane) and the writes (256 pixels x Cp consecutive floats) are
// coalesced.  C <= 32.  grid (ceil(HW/256), B).
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                           int Cp, int dCs, long HW) {
    __shared__ float tile[32][257];
    const int t = threadIdx.x;
    const long p0 = (long)blockIdx.x * 256;
    const long b = blockIdx.y;
    const int np = (int)min((long)256, HW - p0);
    for (int c = 0; c < C; ++c)
        if (t < np) tile[c][t] = src[(b * C + c) * HW + p0 + t];
    __syncthreads();
    float* d = dst + (b * HW + p0) * dCs;
    for (int i = t; i < np * Cp; i += 256) {
        const int pix = i / Cp, c = i - pix * Cp;
        d[(long)pix * dCs + c] = c < C ? tile[c][pix] : 0.f;
    }
}

__global__ __launch_bounds__(256) void add4d_kernel(T4 a, T4 bt, T4 d, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long c = i % d.c; long r = i / d.c;
    const long x = r % d.w; r /= d.w;
    const long y = r % d.h; const long b = r / d.h;
    ((float*)d.p)[b * d.sn + c * d.sc + y * d.sh + x * d.sw] =
        ((const float*)a.p)[b * a.sn + c * a.sc + y * a.sh + x * a.sw] +
        ((const float*)bt.p)[b * bt.sn + c * bt.sc + y * bt.sh + x * bt.sw];
}

// SpaceToDepth(2): dst[b, (2i+j)*C + c, h, w] = x[b, c, 2h+i, 2w+j]; thread per (dst pixel, quadrant, float4)
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ x, int xCs, float* __restrict__ dst, int dCs,
                                                  int Ho, int Wo, int C4, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % C4); long r = i / C4;
    const int ij = (int)(r & 3); r >>= 2;
    const int w = (int)(r % Wo); r /= Wo;
    const int h = (int)(r % Ho);
    const long b = r / Ho;
    const f32x4 v = *(const f32x4*)(x + ((b * (2L * Ho) + 2 * h + (ij >> 1)) * (2L * Wo) + 2 * w + (ij & 1)) * xCs + q * 4);
    *(f32x4*)(dst + ((b * Ho + h) * (long)Wo + w) * dCs + ij * (C4 * 4) + q * 4) = v;
}
// DepthToSpace(2): dst[b, c, 2h+i, 2w+j] = x[b, (2i+j)*C + c, h, w]; C = x.c/4
__global__ __launch_bounds__(256) void d2s_kernel(const float* __restrict__ x, int xCs, float* __restrict__ dst, int dCs,
                                                  int H, int W, int C4, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % C4); long r = i / C4;
    const int ij = (int)(r & 3); r >>= 2;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const long b = r / H;
    const f32x4 v = *(const f32x4*)(x + ((b * H + h) * (long)W + w) * xCs + ij * (C4 * 4) + q * 4);
    *(f32x4*)(dst + ((b * (2L * H) + 2 * h + (ij >> 1)) * (2L * W) + 2 * w + (ij & 1)) * dCs + q * 4) = v;
}

// EdgeConnect input assembly
__global__ __launch_bounds__(256) void ec_inputs_kernel(T4 img, T4 edg, T4 msk, T4 d, int mode, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long x = i % d.w; const long r = i / d.w;
    const long y = r % d.h; const long b = r / d.h;
    const float m = ((const float*)msk.p)[b * msk.sn + y * msk.sh + x * msk.sw];
    const float e = ((const float*)edg.p)[b * edg.sn + y * edg.sh + x * edg.sw];
    const float* ip = (const float*)img.p + b * img.sn + y * img.sh + x * img.sw;
    float* dp = (float*)d.p + b * d.sn + y * d.sh + x * d.sw;
    if (mode == 0) {
        dp[0] = ip[0] * (1.f - m) + m;
        dp[d.sc] = e * (1.f - m);
        dp[2 * d.sc] = m;
        dp[3 * d.sc] = 0.f;                      // channel pad
    } else {
        for (int c = 0; c < 3; ++c) dp[c * d.sc] = ip[c * img.sc] * (1.f - m) + m;
        dp[3 * d.sc] = e;
    }
}

// row-split conv, horizontal gather-sum: one thread per output pixel; t is [B,H,W,Cs] with cout*kw channels
__global__ __launch_bounds__(256) void hshift_sum_kernel(const float* __restrict__ t, int tCs, const float* __restrict__ bias,
                                                         int kw, int pad, int reflect, int act, T4 d, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int W = (int)d.w;
    const int x = (int)(i % W);
    const long row = i / W;                       // b*H + y
    const long y = row % d.h, b = row / d.h;
    float* dp = (float*)d.p + b * d.sn + y * d.sh + (long)x * d.sw;
    for (int co = 0; co < (int)d.c; ++co) {
        float acc = bias[co];
        for (int kx = 0; kx < kw; ++kx) {
            int xx = x + kx - pad;
            bool ok = true;
            if (reflect) xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
            else ok = (unsigned)xx < (unsigned)W;
            if (ok) acc += t[(row * W + xx) * tCs + co * kw + kx];
        }
        dp[co * d.sc] = act1(acc, act);
    }
}

// Same sums for a channel pitch of up to 32 floats (a multiple of 4), staged through LDS: a block owns 256 consecutive
// pixels of one image row, pulls their (+- pad) records in with coalesced 16-byte loads and each thread then picks
// its cout*kw taps out of LDS (odd pitch tCs + 1 floats: conflict-free).  The direct kernel above reads 4 bytes
// per lane at a 128-byte lane stride - every load instruction touches 64 cache lines.
__global__ __launch_bounds__(256) void hshift_sum_lds_kernel(const float* __restrict__ t, int tCs, const float* __restrict__ bias,
                                                             int kw, int pad, int reflect, int act, T4 d, int segs) {
    extern __shared__ float tile[];               // [256 + 2*pad][tCs + 1]
    const int q4 = tCs >> 2, lp = tCs + 1;
    const int W = (int)d.w;
    const long row = blockIdx.x / segs;           // b*H + y
    const int x0 = (blockIdx.x % segs) * 256;
    const int npx = 256 + 2 * pad;
    for (int it = threadIdx.x; it < npx * q4; it += 256) {
        const int p = it / q4, q = it - p * q4;
        const int gx = x0 - pad + p;
        if ((unsigned)gx < (unsigned)W) {
            const float4 v = *(const float4*)(t + ((row * W + gx) * tCs + q * 4));
            float* o = tile + p * lp + q * 4;
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= W) return;
    const long y = row % d.h, b = row / d.h;
    float* dp = (float*)d.p + b * d.sn + y * d.sh + (long)x * d.sw;
    for (int co = 0; co < (int)d.c; ++co) {
        float acc = bias[co];
        for (int kx = 0; kx < kw; ++kx) {
            int xx = x + kx - pad;
            bool ok = true;
            if (reflect) xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
            else ok = (unsigned)xx < (unsigned)W;
            if (ok) acc += tile[(xx - x0 + pad) * lp + co * kw + kx];
        }
        dp[co * d.sc] = act1(acc, act);
    }
}

// first-occurrence argmax over H*W; one block per (b, c)
__global__ __launch_bounds__(256) void argmax_hw_kernel(T4 x, int* __restrict__ idx) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int bc = blockIdx.x;
    const long b = bc / x.c, c = bc % x.c;
    const float* p = (const float*)x.p + b * x.sn + c * x.sc;
    const int HW = (int)(x.h * x.w);
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const int yy = i / (int)x.w, xx = i - yy * (int)x.w;
        const float v = p[yy * x.sh + xx * x.sw];
        if (v > best || bi == 0x7fffffff) { best = v; bi = i; }   // strictly greater keeps the first index
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float ov = sv[threadIdx.x + s]; const int oi = si[threadIdx.x + s];
            if (oi != 0x7fffffff && (si[threadIdx.x] == 0x7fffffff || ov > sv[threadIdx.x] ||
                                     (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))) {
                sv[threadIdx.x] = ov; si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) idx[bc] = si[0];
}

__global__ __launch_bounds__(256) void to_image_u8_kernel(T4 x, T4 d, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long xx = i % d.w; const long r = i / d.w;
    const long y = r % d.h; const long b = r / d.h;
    const float* sp = (const float*)x.p + b * x.sn + y * x.sh + xx * x.sw;
    unsigned char* dp = (unsigned char*)d.p + b * d.sn + y * d.sh + xx * d.sw;
    for (long c = 0; c < d.c; ++c) {
        float v = (sp[c * x.sc] + 1.f) / 2.f * 255.f;          // float32 arithmetic, like numpy on float32
        v = fminf(fmaxf(v, 0.f), 255.f);
        dp[c * d.sc] = (unsigned char)v;                          // truncation
    }
}

__global__ __launch_bounds__(256) void merge_u8_kernel(T4 o, T4 img, T4 msk, T4 d, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long xx = i % d.w; const long r = i / d.w;
    const long y = r % d.h; const long b = r / d.h;
    const float m = ((const float*)msk.p)[b * msk.sn + y * msk.sh + xx * msk.sw];
    const float* op = (const float*)o.p + b * o.sn + y * o.sh + xx * o.sw;
    const float* ip = (const float*)img.p + b * img.sn + y * img.sh + xx * img.sw;
    unsigned char* dp = (unsigned char*)d.p + b * d.sn + y * d.sh + xx * d.sw;
    for (long c = 0; c < d.c; ++c) {
        const float v = (op[c * o.sc] * m + ip[c * img.sc] * (1.f - m)) * 255.f;
        dp[c * d.sc] = (unsigned char)fminf(fmaxf(v, 0.f), 255.f);
    }
}

}  // namespace fusg

using namespace fusg;

static int affine_act_impl(const fusg_tensor* x, const float* scale, const float* shift, int64_t bstride, int32_t act,
                               const fusg_tensor* res, const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(x && dst && is_nhwc(*x) && is_nhwc(*dst) && same_shape(*x, *dst) && x->c % 4 == 0,
               "affine_act: x/dst must be NHWC-physical, same shape, C%%4==0");
    FUSG_CHECK((scale == nullptr) == (shift == nullptr), "affine_act: scale and shift go together");
    if (scale) FUSG_CHECK(((((uintptr_t)scale) | ((uintptr_t)shift)) & 15) == 0 && bstride % 4 == 0, "affine_act: misaligned scale/shift");
    const bool hr = res && res->data;
    if (hr) FUSG_CHECK(is_nhwc(*res) && same_shape(*x, *res), "affine_act: residual must match x");
    FUSG_CHECK(act >= 0 && act <= FUSG_ACT_TANH01, "affine_act: act");
    const int C4 = (int)(x->c / 4);
    const long total = x->n * x->h * x->w * C4;
    unsigned nb = blocks_for(total);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(affine_act_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)x->data, (int)x->sw, scale,
                       shift, (long)bstride, act, hr ? (const float*)res->data : nullptr, hr ? (int)res->sw : 0,
                       (float*)dst->data, (int)dst->sw, (long)(x->h * x->w), C4, total);
    FUSG_LAUNCH_CHECK("affine_act");
    return FUSG_OK;
}
extern "C" int fusg_affine_act(const fusg_tensor* x, const float* scale, const float* shift, int64_t bstride, int32_t act, const fusg_tensor* res, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(affine_act_impl, stream, x, scale, shift, bstride, act, res, dst); }

static int maxpool2_impl(const fusg_tensor* x, const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(x && dst && is_nhwc(*x) && is_nhwc(*dst) && x->c % 4 == 0 && x->c == dst->c && x->n == dst->n &&
               x->h == 2 * dst->h && x->w == 2 * dst->w, "maxpool2: shapes (needs even H,W, NHWC-physical)");
    const int C4 = (int)(x->c / 4);
    const long total = dst->n * dst->h * dst->w * C4;
    unsigned nb = blocks_for(total);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(maxpool2_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)x->data, (int)x->sw,
                       (float*)dst->data, (int)dst->sw, (int)dst->h, (int)dst->w, (int)x->w, C4, total);
    FUSG_LAUNCH_CHECK("maxpool2");
    return FUSG_OK;
}
extern "C" int fusg_maxpool2(const fusg_tensor* x, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(maxpool2_impl, stream, x, dst); }

static int upsample2_add_impl(const fusg_tensor* low, const fusg_tensor* up1, const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(low && up1 && dst && is_nhwc(*low) && is_nhwc(*up1) && is_nhwc(*dst) && same_shape(*up1, *dst) &&
               up1->c % 4 == 0 && low->c == up1->c && low->n == up1->n && up1->h == 2 * low->h && up1->w == 2 * low->w,
               "upsample2_add: shapes");
    const int C4 = (int)(up1->c / 4);
    const long total = dst->n * dst->h * dst->w * C4;
    unsigned nb = blocks_for(total);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(upsample2_add_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)low->data, (int)low->sw,
                       (const float*)up1->data, (int)up1->sw, (float*)dst->data, (int)dst->sw, (int)dst->h, (int)dst->w, C4, total);
    FUSG_LAUNCH_CHECK("upsample2_add");
    return FUSG_OK;
}
extern "C" int fusg_upsample2_add(const fusg_tensor* low, const fusg_tensor* up1, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(upsample2_add_impl, stream, low, up1, dst); }

static int copy4d_impl(const fusg_tensor* src, const fusg_tensor* dst, int32_t dst_c_fill, void* stream) {
    FUSG_CHECK(src && dst && src->data && dst->data && src->dtype == FUSG_F32 && dst->dtype == FUSG_F32, "copy4d: f32 tensors required");
    FUSG_CHECK(same_nhw(*src, *dst) && src->c <= dst->c && dst_c_fill <= dst->c, "copy4d: shape mismatch");
    const long HW = src->h * src->w;
    const bool src_nchw = src->sw == 1 && src->sh == src->w && src->sc == HW && src->sn == src->c * HW;
    if (src_nchw && is_nhwc(*dst) && src->c <= 32 && dst_c_fill <= 32 && dst_c_fill <= dst->sw) {
        const int cp = dst_c_fill > src->c ? dst_c_fill : (int)src->c;
        hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)src->n), dim3(256), 0,
                           (hipStream_t)stream, (const float*)src->data, (float*)dst->data, (int)src->c, cp, (int)dst->sw, HW);
        FUSG_LAUNCH_CHECK("nchw_to_nhwc");
        return FUSG_OK;
    }
    const long total = src->n * src->h * src->w;
    hipLaunchKernelGGL(copy4d_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, view(*src), view(*dst),
                       (int)dst_c_fill, total);
    FUSG_LAUNCH_CHECK("copy4d");
    return FUSG_OK;
}
extern "C" int fusg_copy4d(const fusg_tensor* src, const fusg_tensor* dst, int32_t dst_c_fill, void* stream) { return fusg::plan_dispatch(copy4d_impl, stream, src, dst, dst_c_fill); }

static int add4d_impl(const fusg_tensor* a, const fusg_tensor* b, const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(a && b && dst && a->data && b->data && dst->data && same_shape(*a, *b) && same_shape(*a, *dst) &&
               a->dtype == FUSG_F32 && b->dtype == FUSG_F32 && dst->dtype == FUSG_F32, "add4d: shape/dtype mismatch");
    const long total = dst->n * dst->h * dst->w * dst->c;
    hipLaunchKernelGGL(add4d_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, view(*a), view(*b), view(*dst), total);
    FUSG_LAUNCH_CHECK("add4d");
    return FUSG_OK;
}
extern "C" int fusg_add4d(const fusg_tensor* a, const fusg_tensor* b, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(add4d_impl, stream, a, b, dst); }

static int space_to_depth2_impl(const fusg_tensor* x, const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(x && dst && is_nhwc(*x) && is_nhwc(*dst) && x->c % 4 == 0 && dst->c == 4 * x->c && x->h == 2 * dst->h &&
               x->w == 2 * dst->w && x->n == dst->n, "space_to_depth2: shapes");
    const int C4 = (int)(x->c / 4);
    const long total = dst->n * dst->h * dst->w * 4 * C4;
    hipLaunchKernelGGL(s2d_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x->data, (int)x->sw,
                       (float*)dst->data, (int)dst->sw, (int)dst->h, (int)dst->w, C4, total);
    FUSG_LAUNCH_CHECK("space_to_depth2");
    return FUSG_OK;
}
extern "C" int fusg_space_to_depth2(const fusg_tensor* x, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(space_to_depth2_impl, stream, x, dst); }

static int depth_to_space2_impl(const fusg_tensor* x, const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(x && dst && is_nhwc(*x) && is_nhwc(*dst) && x->c % 16 == 0 && x->c == 4 * dst->c && dst->h == 2 * x->h &&
               dst->w == 2 * x->w && x->n == dst->n, "depth_to_space2: shapes");
    const int C4 = (int)(dst->c / 4);
    const long total = x->n * x->h * x->w * 4 * C4;
    hipLaunchKernelGGL(d2s_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x->data, (int)x->sw,
                       (float*)dst->data, (int)dst->sw, (int)x->h, (int)x->w, C4, total);
    FUSG_LAUNCH_CHECK("depth_to_space2");
    return FUSG_OK;
}
extern "C" int fusg_depth_to_space2(const fusg_tensor* x, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(depth_to_space2_impl, stream, x, dst); }

static int ec_inputs_impl(const fusg_tensor* images, const fusg_tensor* edges, const fusg_tensor* masks,
                              const fusg_tensor* dst, int32_t mode, void* stream) {
    FUSG_CHECK(images && edges && masks && dst && images->data && edges->data && masks->data && dst->data, "ec_inputs: null tensor");
    FUSG_CHECK(mode == 0 || mode == 1, "ec_inputs: mode");
    FUSG_CHECK(images->c == (mode == 0 ? 1 : 3) && edges->c == 1 && masks->c == 1 && dst->c >= 4 && same_nhw(*images, *dst) &&
               same_nhw(*edges, *dst) && same_nhw(*masks, *dst), "ec_inputs: shapes");
    const long total = dst->n * dst->h * dst->w;
    hipLaunchKernelGGL(ec_inputs_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, view(*images), view(*edges),
                       view(*masks), view(*dst), mode, total);
    FUSG_LAUNCH_CHECK("ec_inputs");
    return FUSG_OK;
}
extern "C" int fusg_ec_inputs(const fusg_tensor* images, const fusg_tensor* edges, const fusg_tensor* masks, const fusg_tensor* dst, int32_t mode, void* stream) { return fusg::plan_dispatch(ec_inputs_impl, stream, images, edges, masks, dst, mode); }

static int hshift_sum_impl(const fusg_tensor* t, const float* bias, int32_t kw, int32_t pad, int32_t pad_mode, int32_t act,
                               const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(t && dst && bias && is_nhwc(*t) && dst->data && dst->dtype == FUSG_F32 && same_nhw(*t, *dst), "hshift_sum: tensors");
    FUSG_CHECK(kw >= 1 && kw <= 15 && pad >= 0 && pad < dst->w && dst->c * kw <= t->c, "hshift_sum: kw %d pad %d cout %ld tc %ld", kw, pad, (long)dst->c, (long)t->c);
    FUSG_CHECK(act >= 0 && act <= FUSG_ACT_TANH01 && (pad_mode == 0 || pad_mode == 1), "hshift_sum: act/pad_mode");
    const long total = dst->n * dst->h * dst->w;
    if (t->sw % 4 == 0 && t->sw <= 32 && pad <= 16 && dst->w > 2 * pad && (((uintptr_t)t->data) & 15) == 0) {
        const int segs = (int)((dst->w + 255) / 256);
        const long nblk = dst->n * dst->h * segs;
        FUSG_CHECK(nblk < (1L << 31), "hshift_sum: grid too large");
        hipLaunchKernelGGL(hshift_sum_lds_kernel, dim3((unsigned)nblk), dim3(256),
                           (size_t)(256 + 2 * pad) * (t->sw + 1) * sizeof(float), (hipStream_t)stream, (const float*)t->data, (int)t->sw, bias, kw, pad, pad_mode, act, view(*dst), segs);
        FUSG_LAUNCH_CHECK("hshift_sum");
        return FUSG_OK;
    }
    hipLaunchKernelGGL(hshift_sum_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)t->data,
                       (int)t->sw, bias, kw, pad, pad_mode, act, view(*dst), total);
    FUSG_LAUNCH_CHECK("hshift_sum");
    return FUSG_OK;
}
extern "C" int fusg_hshift_sum(const fusg_tensor* t, const float* bias, int32_t kw, int32_t pad, int32_t pad_mode, int32_t act, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(hshift_sum_impl, stream, t, bias, kw, pad, pad_mode, act, dst); }

static int argmax_hw_impl(const fusg_tensor* x, int32_t* idx, void* stream) {
    FUSG_CHECK(x && x->data && idx && x->dtype == FUSG_F32 && x->h * x->w >= 1 && x->h * x->w < (1L << 30), "argmax_hw: bad tensor");
    hipLaunchKernelGGL(argmax_hw_kernel, dim3((unsigned)(x->n * x->c)), dim3(256), 0, (hipStream_t)stream, view(*x), idx);
    FUSG_LAUNCH_CHECK("argmax_hw");
    return FUSG_OK;
}
extern "C" int fusg_argmax_hw(const fusg_tensor* x, int32_t* idx, void* stream) { return fusg::plan_dispatch(argmax_hw_impl, stream, x, idx); }

static int to_image_u8_impl(const fusg_tensor* x, const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(x && dst && x->data && dst->data && x->dtype == FUSG_F32 && dst->dtype == FUSG_U8 && same_shape(*x, *dst),
               "to_image_u8: shape/dtype");
    const long total = dst->n * dst->h * dst->w;
    hipLaunchKernelGGL(to_image_u8_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, view(*x), view(*dst), total);
    FUSG_LAUNCH_CHECK("to_image_u8");
    return FUSG_OK;
}
extern "C" int fusg_to_image_u8(const fusg_tensor* x, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(to_image_u8_impl, stream, x, dst); }

static int merge_u8_impl(const fusg_tensor* out, const fusg_tensor* img, const fusg_tensor* mask, const fusg_tensor* dst,
                             void* stream) {
    FUSG_CHECK(out && img && mask && dst && out->data && img->data && mask->data && dst->data && dst->dtype == FUSG_U8 &&
               same_shape(*out, *img) && same_shape(*out, *dst) && same_nhw(*mask, *dst) && mask->c == 1, "merge_u8: shapes");
    const long total = dst->n * dst->h * dst->w;
    hipLaunchKernelGGL(merge_u8_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, view(*out), view(*img),
                       view(*mask), view(*dst), total);
    FUSG_LAUNCH_CHECK("merge_u8");
    return FUSG_OK;
}
extern "C" int fusg_merge_u8(const fusg_tensor* out, const fusg_tensor* img, const fusg_tensor* mask, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(merge_u8_impl, stream, out, img, mask, dst); }
