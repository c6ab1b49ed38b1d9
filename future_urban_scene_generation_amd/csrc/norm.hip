// Normalisation statistics for gfx950: InstanceNorm2d (per (b,c) over H*W) and the ICN's custom
// whole-sample LayerNorm (per b over C*H*W, unbiased std, eps on std).  HBM-bound streaming
// reductions over NHWC activations: float4 per lane along channels (coalesced 16 B/lane), pixels
// split over lanes and over `nchunk` workgroups per sample, LDS tree across pixel lanes, and a
// deterministic second stage in fp64 (no atomics: results are run-to-run reproducible).
// Sums are shifted by a per-(b,c) pivot p = x[b,0,0,c] so that E[(x-p)^2] - E[x-p]^2 does not cancel.
#include "common.h"

namespace fusg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid (nchunk, B, ceil(C4/64)); block 256 = 64 channel-quads x 4 pixel lanes (or fewer quads, more lanes)
__global__ __launch_bounds__(256) void chan_stats_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                         int HW, int C, int Cs, int nchunk) {
    __shared__ f32x4 red[2][256];
    const int C4 = C >> 2;
    const int cg0 = blockIdx.z * 64;
    const int nq = min(64, C4 - cg0);              // channel quads handled by this block
    // lanes per quad: largest power of two with nq * pl <= 256
    int pl = 256 / nq;
    pl = 1 << (31 - __clz(pl));
    const int t = threadIdx.x;
    const int q = t % nq;
    const int pj = t / nq;
    const int b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int per = (HW + nchunk - 1) / nchunk;
    const int p0 = chunk * per;
    const int p1 = min(HW, p0 + per);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (pj < pl) {
        const float* xb = x + (long)b * HW * Cs + (cg0 + q) * 4;
        const f32x4 piv = *(const f32x4*)xb;
        for (int p = p0 + pj; p < p1; p += pl) {
            f32x4 v = *(const f32x4*)(xb + (long)p * Cs);
            v -= piv;
            s1 += v;
            s2 += v * v;
        }
    }
    red[0][t] = s1;
    red[1][t] = s2;
    __syncthreads();
    if (pj == 0) {
        for (int j = 1; j < pl; ++j) { s1 += red[0][j * nq + q]; s2 += red[1][j * nq + q]; }
        float* o = partial + (((long)b * nchunk + chunk) * C + (cg0 + q) * 4) * 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) { o[2 * c] = s1[c]; o[2 * c + 1] = s2[c]; }
    }
}

// one thread per (b, c)
__global__ void in_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial, int B, int HW, int C,
                                   int Cs, int nchunk, float eps, float* __restrict__ scale, float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nchunk; ++k) {
        const float* o = partial + (((long)b * nchunk + k) * C + c) * 2;
        s1 += (double)o[0];
        s2 += (double)o[1];
    }
    const double n = (double)HW;
    const double ms = s1 / n;
    double var = s2 / n - ms * ms;                // biased (F.instance_norm)
    if (var < 0.0) var = 0.0;
    const double mean = (double)x[(long)b * HW * Cs + c] + ms;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    scale[i] = (float)rstd;
    shift[i] = (float)(-mean * rstd);
}

// one block per sample b, 256 threads striding channels; combine channel moments (Chan et al.) in fp64
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                          int HW, int C, int Cs, int nchunk, float eps,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double red[256];
    __shared__ double bc[2];
    const int b = blockIdx.x;
    const int t = threadIdx.x;
    const double n = (double)HW;
    // pass 1: total sum
    double tsum = 0.0;
    for (int c = t; c < C; c += 256) {
        double s1 = 0.0;
        for (int k = 0; k < nchunk; ++k) s1 += (double)partial[(((long)b * nchunk + k) * C + c) * 2];
        tsum += s1 + n * (double)x[(long)b * HW * Cs + c];
    }
    red[t] = tsum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
    if (t == 0) bc[0] = red[0] / (n * C);
    __syncthreads();
    const double mean = bc[0];
    // pass 2: total M2 = sum_c [ M2_c + n (mean_c - mean)^2 ]
    double m2 = 0.0;
    for (int c = t; c < C; c += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < nchunk; ++k) {
            const float* o = partial + (((long)b * nchunk + k) * C + c) * 2;
            s1 += (double)o[0];
            s2 += (double)o[1];
        }
        const double mc = (double)x[(long)b * HW * Cs + c] + s1 / n;
        double m2c = s2 - s1 * s1 / n;
        if (m2c < 0.0) m2c = 0.0;
        m2 += m2c + n * (mc - mean) * (mc - mean);
    }
    __syncthreads();
    red[t] = m2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
    if (t == 0) bc[1] = sqrt(red[0] / (n * C - 1.0));      // unbiased std (torch.std default)
    __syncthreads();
    const double inv = 1.0 / (bc[1] + (double)eps);         // eps added to the std (models.py:30)
    for (int c = t; c < C; c += 256) {
        const double sc = (double)gamma[c] * inv;
        scale[(long)b * C + c] = (float)sc;
        shift[(long)b * C + c] = (float)((double)beta[c] - mean * sc);
    }
}

// ---- finalisation from the slot statistics fused into the conv epilogue: slots [B][nslots][C][2] =
// (mean, centred M2) of 32 pixels each.  Equal counts, so mean = avg(mean_s), M2 = sum M2_s + 32 sum (mean_s-mean)^2.
// Block = 16 channels x 16 slot lanes (128 B contiguous per slot row); grid (ceil(C/16), B).  One pass in fp64:
// sum of slot means, of their squares and of the slot M2s, then an LDS tree over the slot lanes.
__global__ __launch_bounds__(256) void in_finalize_slots_kernel(const float* __restrict__ slots, int nslots, int C,
                                                                float eps, float* __restrict__ scale,
                                                                float* __restrict__ shift) {
    __shared__ double red[3][256];
    const int t = threadIdx.x;
    const int cl = t & 15, sl = t >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int b = blockIdx.y;
    double sm = 0.0, sq = 0.0, s2 = 0.0;
    if (c < C) {
        const float2* s = (const float2*)slots + (long)b * nslots * C + c;
#pragma unroll 4
        for (int k = sl; k < nslots; k += 16) {
            const float2 v = s[(long)k * C];
            sm += (double)v.x;
            sq += (double)v.x * (double)v.x;
            s2 += (double)v.y;
        }
    }
    red[0][t] = sm; red[1][t] = sq; red[2][t] = s2;
    __syncthreads();
    for (int w = 128; w >= 16; w >>= 1) {
        if (t < w) { red[0][t] += red[0][t + w]; red[1][t] += red[1][t + w]; red[2][t] += red[2][t + w]; }
        __syncthreads();
    }
    if (t < 16 && c < C) {
        const double n = (double)nslots;
        const double mean = red[0][t] / n;
        double m2 = red[2][t] + 32.0 * (red[1][t] - red[0][t] * mean);
        if (m2 < 0.0) m2 = 0.0;
        const double var = m2 / (32.0 * n);                // biased (F.instance_norm)
        const double rstd = 1.0 / sqrt(var + (double)eps);
        scale[(long)b * C + c] = (float)rstd;
        shift[(long)b * C + c] = (float)(-mean * rstd);
    }
}

// one block of 1024 threads per sample; same one-pass fp64 moments over all (slot, channel) pairs
__global__ __launch_bounds__(1024) void ln_finalize_slots_kernel(const float* __restrict__ slots, int nslots, int C, float eps,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double red[3][1024];
    const int b = blockIdx.x, t = threadIdx.x;
    const long total = (long)nslots * C;                  // (slot, channel) pairs, 32 pixels each
    const float2* s = (const float2*)slots + (long)b * total;
    double sm = 0.0, sq = 0.0, s2 = 0.0;
#pragma unroll 4
    for (long k = t; k < total; k += 1024) {
        const float2 v = s[k];
        sm += (double)v.x;
        sq += (double)v.x * (double)v.x;
        s2 += (double)v.y;
    }
    red[0][t] = sm; red[1][t] = sq; red[2][t] = s2;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (t < w) { red[0][t] += red[0][t + w]; red[1][t] += red[1][t + w]; red[2][t] += red[2][t + w]; }
        __syncthreads();
    }
    const double mean = red[0][0] / (double)total;
    double m2 = red[2][0] + 32.0 * (red[1][0] - red[0][0] * mean);
    if (m2 < 0.0) m2 = 0.0;
    const double sd = sqrt(m2 / (32.0 * (double)total - 1.0));             // unbiased std
    const double inv = 1.0 / (sd + (double)eps);
    for (int c = t; c < C; c += 1024) {
        const double sc = (double)gamma[c] * inv;
        scale[(long)b * C + c] = (float)sc;
        shift[(long)b * C + c] = (float)((double)beta[c] - mean * sc);
    }
}

}  // namespace fusg

using namespace fusg;

static int in_finalize_slots_impl(const float* slots, int32_t batch, int32_t nslots, int32_t channels, float eps,
                                      float* scale, float* shift, void* stream) {
    FUSG_CHECK(slots && scale && shift && batch >= 1 && nslots >= 1 && channels >= 1, "in_finalize_slots: bad arguments");
    hipLaunchKernelGGL(in_finalize_slots_kernel, dim3((channels + 15) / 16, batch), dim3(256), 0, (hipStream_t)stream, slots,
                       nslots, channels, eps, scale, shift);
    FUSG_LAUNCH_CHECK("in_finalize_slots");
    return FUSG_OK;
}
extern "C" int fusg_in_finalize_slots(const float* slots, int32_t batch, int32_t nslots, int32_t channels, float eps, float* scale, float* shift, void* stream) { return fusg::plan_dispatch(in_finalize_slots_impl, stream, slots, batch, nslots, channels, eps, scale, shift); }

static int ln_finalize_slots_impl(const float* slots, int32_t batch, int32_t nslots, int32_t channels, float eps,
                                      const float* gamma, const float* beta, float* scale, float* shift, void* stream) {
    FUSG_CHECK(slots && scale && shift && gamma && beta && batch >= 1 && nslots >= 1 && channels >= 1 &&
               (long)nslots * channels * 32 > 1, "ln_finalize_slots: bad arguments");
    hipLaunchKernelGGL(ln_finalize_slots_kernel, dim3((unsigned)batch), dim3(1024), 0, (hipStream_t)stream, slots, nslots,
                       channels, eps, gamma, beta, scale, shift);
    FUSG_LAUNCH_CHECK("ln_finalize_slots");
    return FUSG_OK;
}
extern "C" int fusg_ln_finalize_slots(const float* slots, int32_t batch, int32_t nslots, int32_t channels, float eps, const float* gamma, const float* beta, float* scale, float* shift, void* stream) { return fusg::plan_dispatch(ln_finalize_slots_impl, stream, slots, batch, nslots, channels, eps, gamma, beta, scale, shift); }

static int chan_stats_impl(const fusg_tensor* x, float* partial, int32_t nchunk, void* stream) {
    FUSG_CHECK(x && is_nhwc(*x) && x->c % 4 == 0, "chan_stats: x must be NHWC-physical with C%%4==0");
    FUSG_CHECK(partial && nchunk >= 1 && nchunk <= 4096, "chan_stats: bad partial/nchunk");
    const long HW = x->h * x->w;
    FUSG_CHECK(HW >= 1 && HW < (1L << 30), "chan_stats: HW out of range");
    const int C4 = (int)(x->c / 4);
    dim3 grid(nchunk, (unsigned)x->n, (C4 + 63) / 64);
    hipLaunchKernelGGL(chan_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x->data, partial,
                       (int)HW, (int)x->c, (int)x->sw, nchunk);
    FUSG_LAUNCH_CHECK("chan_stats");
    return FUSG_OK;
}
extern "C" int fusg_chan_stats(const fusg_tensor* x, float* partial, int32_t nchunk, void* stream) { return fusg::plan_dispatch(chan_stats_impl, stream, x, partial, nchunk); }

static int in_finalize_impl(const fusg_tensor* x, const float* partial, int32_t nchunk, float eps, float* scale,
                                float* shift, void* stream) {
    FUSG_CHECK(x && is_nhwc(*x) && partial && scale && shift && nchunk >= 1, "in_finalize: bad arguments");
    const int total = (int)(x->n * x->c);
    hipLaunchKernelGGL(in_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x->data, partial, (int)x->n, (int)(x->h * x->w), (int)x->c, (int)x->sw, nchunk, eps,
                       scale, shift);
    FUSG_LAUNCH_CHECK("in_finalize");
    return FUSG_OK;
}
extern "C" int fusg_in_finalize(const fusg_tensor* x, const float* partial, int32_t nchunk, float eps, float* scale, float* shift, void* stream) { return fusg::plan_dispatch(in_finalize_impl, stream, x, partial, nchunk, eps, scale, shift); }

static int ln_finalize_impl(const fusg_tensor* x, const float* partial, int32_t nchunk, float eps, const float* gamma,
                                const float* beta, float* scale, float* shift, void* stream) {
    FUSG_CHECK(x && is_nhwc(*x) && partial && scale && shift && gamma && beta && nchunk >= 1, "ln_finalize: bad arguments");
    FUSG_CHECK(x->c * x->h * x->w > 1, "ln_finalize: needs more than one element per sample");
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)x->n), dim3(256), 0, (hipStream_t)stream, (const float*)x->data,
                       partial, (int)(x->h * x->w), (int)x->c, (int)x->sw, nchunk, eps, gamma, beta, scale, shift);
    FUSG_LAUNCH_CHECK("ln_finalize");
    return FUSG_OK;
}
extern "C" int fusg_ln_finalize(const fusg_tensor* x, const float* partial, int32_t nchunk, float eps, const float* gamma, const float* beta, float* scale, float* shift, void* stream) { return fusg::plan_dispatch(ln_finalize_impl, stream, x, partial, nchunk, eps, gamma, beta, scale, shift); }
