// Small-spatial split-fp16 convolution: the layers whose images are 16 x 16 pixels or smaller (the VUnet's bottleneck levels
// and AutoRegressiveBlocks, vunet/models.py:17-89; the hourglass's low levels) - M = B * Ho * Wo of a few hundred rows, K up
// to 9216.  Nothing about them is bound by a pipe: they are chains of dependent launches, and what a launch costs is its
// LATENCY.  The generic gather (conv_kernel_h3.h) walks K in 32-wide steps with a barrier and a two-deep prefetch per step
// and needs a second launch to sum its split-K slabs: 13 - 17 us per layer + 6 us of reduce (profiles/r03_*), at an
// MFMA-pipe busy of 0.12.  This kernel is built for latency instead:
//
//   * a workgroup owns 32 output rows (two 16-row MFMA tiles: half an 8 x 8 image, two 4 x 4 images, eight 2 x 2 images)
//     x 32 output channels x a RANGE OF 32-CHANNEL CHUNKS (all of K when it fits, else gridDim.z ranges + the existing
//     deterministic slab reduce);
//   * its four waves split the (chunk, tap) steps round-robin and EVERY wave issues the loads of ALL its weight fragments
//     (fragment-order copy of the halo kernel: one contiguous 1 KiB wave load per fragment, straight to VGPRs, up to
//     9 steps = 36 KiB per wave) before anything else - one memory round trip for the whole weight stream instead of one
//     per K-step;
//   * the input region of the 32 rows (whole small images, or the rows of one image, with their zero padding
//     materialised) is gathered, pre-processed (ELU / ReLU / affine) and split into fp16 (hi, lo') ONCE into LDS, all the
//     workgroup's channels at a time - every tap of every wave then reads its A fragments from that image (as in the
//     halo kernel), so a 3 x 3 layer converts each activation once, not nine times;
//   * no barrier inside the contraction: one after the staging, one before the four partial tiles are summed in LDS (in
//     wave order: deterministic) by all 256 threads, which also run the epilogue (bias, activation, residuals, DepthToSpace /
//     SpaceToDepth / strided stores: the same mapping functions as every other kernel).
//
// Geometry handled: stride 1 or 2, k x k with k <= 3 (dense tap grid, dilation 1), zero padding, sources with a multiple of
// 32 channels, Ho * Wo a divisor or a multiple of 32 with Wo | 32.  Everything else keeps the other kernels.
#pragma once
#include "conv_kernel_h3.h"

namespace fusg {

constexpr int SMALL_NS = 9;            // weight steps a wave holds in registers at once (4 fragments of 4 VGPRs each)
constexpr int SMALL_ROWS = 32;         // output rows per workgroup
constexpr int SMALL_MAXPIX = 192;      // staged input pixels per workgroup (host-checked; one thread per pixel fills the source table)

struct SmallK {
    ConvK c;
    int kh, kw, pad_h, pad_w, stride;
    int nch0, nch32;            // 32-channel chunks of src0 / of both sources
    int ntaps;
    unsigned long long tapslab; // 4 bits per tap ky * kw + kx: its slab (first index of the fragment-order weights).  Packed into a
                                // scalar on purpose: a dynamically indexed ARRAY in the kernel arguments is read with vector loads
                                // from the kernarg segment - host memory - at ~4 us per dependent load (measured: every launch
                                // of the first version took 7.9 us against 3.7 us for a trivial kernel)
    const _Float16* wfrag;      // [slab][chunk][cout_pad/32][16-column half][hi|lo][64 lanes][8] halves
    int nt32;
    int nchw;                   // chunks per K range (gridDim.z ranges)
    int nimg;                   // images a workgroup's 32 rows span (1: a run of pixels of ONE image)
    int rpi;                    // nimg > 1: output pixels per image (= Ho * Wo, a power of two)
    int tpi;                    // nimg == 1: 32-row tiles per image = ceil(Ho * Wo / 32) (the last one may be ragged)
    int wide;                   // nimg == 1: 1 = a tile is a 32-pixel segment of one output row (Wo >= 32), 0 = 32 / Wo whole rows
    int pad_mode;               // FUSG_PAD_ZERO or FUSG_PAD_REPLICATE (clamped coordinates: the ring windows of the ICN's up-convolutions)
    int RIN, WIN, NPIX;         // staged region per image (rows, columns), pixels in all = nimg * RIN * WIN
    int rpi_shift;              // log2(rpi)
    unsigned m_wo, m_hw, m_win, m_rw, m_npix, m_taps, m_tpi;   // reciprocals (sdiv) of Wo, Ho*Wo, WIN, RIN*WIN, NPIX, ntaps, tpi: every dividend is < 2^16
    int part_off;               // byte offset of the partial tiles in dynamic LDS
};

__device__ __forceinline__ int sdiv(int n, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n; }

template <int PK>
__global__ __launch_bounds__(256, 1) void conv_small_h3(const SmallK sk) {
    const ConvK& p = sk.c;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_s[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);     // scalar: the step bookkeeping below stays on the SALU
    const int tile = blockIdx.x;
    const int mt = tile / p.NT, nt = tile - mt * p.NT;
    const int ks = blockIdx.z;
    const int hw = p.Ho * p.Wo;
    // the workgroup's 32 rows: nimg == 1: pixels p0 .. p0 + 31 of image b0's (window of the) output, row-major - a segment of
    // one row (wide) or 32 / Wo whole rows; nimg > 1: the whole outputs of images b0 .. b0 + nimg - 1
    int b0, p0 = 0, oy_t = 0, ox_t = 0;
    if (sk.nimg == 1) {
        b0 = sdiv(mt, sk.m_tpi);
        p0 = (mt - b0 * sk.tpi) * SMALL_ROWS;
        oy_t = sdiv(p0, sk.m_wo);
        ox_t = sk.wide ? p0 - oy_t * p.Wo : 0;
    } else {
        b0 = mt * sk.nimg;
    }
    const int cg0 = ks * sk.nchw;
    const int nmine = min(sk.nch32 - cg0, sk.nchw);            // >= 1 (host: every range is non-empty)
    const int nsteps = nmine * sk.ntaps;
    const int NPIX = sk.NPIX;
    _Float16* Ah = smem_s;                                       // [chunk][pixel][32 halves], 16-byte slots swizzled by pixel
    _Float16* Al = Ah + nmine * NPIX * 32;
    int* pixsrc = (int*)((char*)smem_s + sk.part_off);           // [NPIX] source pixel (b*H*W + iy*W + ix) or -1; the partial tiles reuse it
    float* part = (float*)((char*)smem_s + sk.part_off);         // [4 waves][32 rows][32 columns]

    // ---- this wave's weight fragments: every load of the first round is issued before anything else
    const _Float16* wbase = sk.wfrag + ((long)nt * 4 * 64 + lane) * 8;
    const long wchunk = (long)sk.nt32 * 4 * 64 * 8;              // halves per (slab, chunk)
    const long wslab = (long)sk.nch32 * wchunk;                  // halves per slab (tap)
    struct BSet { h8 f[SMALL_NS][2][2]; };                       // [step][16-column half][hi, lo]
    BSet bw;
    auto step_of = [&](int tt, int& cl, int& tap) __attribute__((always_inline)) {      // step -> (chunk, tap): chunk-major
        cl = sdiv(tt, sk.m_taps);
        tap = tt - cl * sk.ntaps;
    };
    auto b_issue = [&](int step0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SMALL_NS; ++i) {
            const int tt = step0 + wave + 4 * i;
            if (tt < nsteps) {
                int cl, tap;
                step_of(tt, cl, tap);
                const int slab = (int)((sk.tapslab >> (4 * tap)) & 15ull);
                const _Float16* w = wbase + (long)slab * wslab + (long)(cg0 + cl) * wchunk;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) bw.f[i][ct][hl] = *(const h8*)(w + (ct * 2 + hl) * 512);
            }
        }
    };
    b_issue(0);

    // ---- source pixel of every staged pixel (one thread per pixel: the divisions happen once)
    if (t < NPIX) {
        const int j = sdiv(t, sk.m_rw), q = t - j * (sk.RIN * sk.WIN);
        const int ry = sdiv(q, sk.m_win), rx = q - ry * sk.WIN;
        const int b = b0 + j;
        int iy = (oy_t + p.qy0) * sk.stride - sk.pad_h + ry, ix = (ox_t + p.qx0) * sk.stride - sk.pad_w + rx;
        bool ok = b < p.B;
        if (sk.pad_mode == FUSG_PAD_REPLICATE) { iy = min(max(iy, 0), p.H - 1); ix = min(max(ix, 0), p.W - 1); }
        else ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        pixsrc[t] = ok ? (b * p.H + iy) * p.W + ix : -1;
    }
    __syncthreads();

    // ---- stage the input region: gather -> pre-op -> fp16 split -> LDS, every chunk of this K range
    const float vfloor = (PK != PK_ELU && p.pre_relu) ? 0.f : -__builtin_inff();
    float amax = 0.f;
    {
        const int kc = t & 7;
        const int nitems = nmine * NPIX * 8;
        constexpr int UN = 4;
        for (int it0 = t; it0 < nitems; it0 += 256 * UN) {
            f32x4 v[UN];
            int lo_[UN], srcp[UN], cidx[UN], pxs[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int it = it0 + 256 * u;
                lo_[u] = -1;
                if (it < nitems) {
                    const int q = it >> 3;
                    const int c = sdiv(q, sk.m_npix), px = q - c * NPIX;
                    const int cg = cg0 + c;
                    const bool s1 = cg >= sk.nch0;
                    const int sp = pixsrc[px];
                    srcp[u] = sp;
                    pxs[u] = px;
                    cidx[u] = (s1 ? p.C0 + (cg - sk.nch0) * 32 : cg * 32) + kc * 4;
                    const float* base = s1 ? p.src1 + (long)max(sp, 0) * p.Cs1 + (cg - sk.nch0) * 32 + kc * 4
                                           : p.src0 + (long)max(sp, 0) * p.Cs0 + cg * 32 + kc * 4;
                    v[u] = *(const f32x4*)(sp >= 0 ? base : p.zeros);
                    lo_[u] = (c * NPIX + px) * 32 + ((((kc >> 1) ^ (((px >> 2) & 1) << 1)) << 3) | ((kc & 1) << 2));
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (lo_[u] < 0) continue;
                f32x4 x = v[u];
                if (PK == PK_ELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[c] = elu1(x[c]);
                } else if (PK == PK_AFFINE) {
                    const int b = min(b0 + sdiv(pxs[u], sk.m_rw), p.B - 1);       // the staged pixel's image (no division by H * W)
                    const f32x4 sc = *(const f32x4*)(p.pre_scale + (long)b * p.pre_bstride + cidx[u]);
                    const f32x4 sh = *(const f32x4*)(p.pre_shift + (long)b * p.pre_bstride + cidx[u]);
                    const bool ok = srcp[u] >= 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const float y = fmaf(x[c], sc[c], sh[c]); x[c] = ok ? y : 0.f; }
                }
                h4 hi, lo;
                if constexpr (PK == PK_ELU) split4<false>(x, vfloor, hi, lo, amax); else split4(x, vfloor, hi, lo, amax);
                *(h4*)(Ah + lo_[u]) = hi;
                *(h4*)(Al + lo_[u]) = lo;
            }
        }
    }

    // ---- this lane's A rows: output row 16 f + (lane & 15) of the workgroup's 32 -> its pixel in the staged region
    int abase[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int r = f * 16 + (lane & 15);
        int j = 0, rem = r;
        if (sk.nimg > 1) { j = min(r >> sk.rpi_shift, sk.nimg - 1); rem = r & (sk.rpi - 1); }      // (rows past the last image: masked below)
        int oyl = 0, ox = rem;
        if (!(sk.nimg == 1 && sk.wide)) { oyl = sdiv(rem, sk.m_wo); ox = rem - oyl * p.Wo; }
        abase[f] = (j * sk.RIN + oyl * sk.stride) * sk.WIN + ox * sk.stride;
    }
    __syncthreads();                                             // the staged image is complete (and pixsrc is dead)

    f32x4 acc[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[f][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- the epilogue's operands (bias, weight scale, residuals of this thread's (row, 4 columns)) are fetched NOW, so that
    // their round trip overlaps the contraction instead of following it
    const int row = t >> 3, c4 = t & 7;
    const int n = nt * 32 + c4 * 4;
    int m;                                                       // this thread's output row in the launch's M = B * Ho * Wo, or -1
    if (sk.nimg == 1) m = p0 + row < hw ? b0 * hw + p0 + row : -1;
    else { const int j = row >> sk.rpi_shift; m = (j < sk.nimg && b0 + j < p.B) ? (b0 + j) * hw + (row & (sk.rpi - 1)) : -1; }
    const bool live = m >= 0 && n < p.Cout && p.ksplit <= 1;
    PixOff po, co;
    po.d = po.r0 = po.r1 = 0;
    co.d = co.r0 = co.r1 = 0;
    f32x4 bs4 = {0.f, 0.f, 0.f, 0.f}, wsc = {1.f, 1.f, 1.f, 1.f}, rv0 = {0.f, 0.f, 0.f, 0.f}, rv1 = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        pix_offsets(p, 0, m, po);
        chan_offsets(p, n, co);
        if (p.vec_epi) {
            bs4 = *(const f32x4*)(p.bias + n);
            if (p.wscale) wsc = *(const f32x4*)(p.wscale + n);
            if (p.res0) rv0 = *(const f32x4*)(p.res0 + po.r0 + co.r0);
            if (p.res1) rv1 = *(const f32x4*)(p.res1 + po.r1 + co.r1);
        }
    }

    for (int step0 = 0; step0 < nsteps; step0 += 4 * SMALL_NS) {
        if (step0 > 0) b_issue(step0);                           // a later round of a long K range
#pragma unroll
        for (int i = 0; i < SMALL_NS; ++i) {
            const int tt = step0 + wave + 4 * i;
            if (tt < nsteps) {
                int cl, tap;
                step_of(tt, cl, tap);
                const int ky = sk.kw == 1 ? tap : (tap * 11) >> 5, kx = tap - ky * sk.kw;       // tap / 3 for tap < 16
                h8 ah[2], al[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int px = abase[f] + ky * sk.WIN + kx;
                    const int off = (cl * NPIX + px) * 32 + (((lane >> 4) ^ (((px >> 2) & 1) << 1)) << 3);
                    ah[f] = *(const h8*)(Ah + off);
                    al[f] = *(const h8*)(Al + off);
                }
                h8 bs[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) bs[ct] = scale_m11(bw.f[i][ct][0]);
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            acc[f][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                term == 2 ? al[f] : ah[f], term == 0 ? bw.f[i][ct][0] : term == 1 ? bw.f[i][ct][1] : bs[ct],
                                acc[f][ct], 0, 0, 0);
            }
        }
    }
    report_range(p, amax);

    // ---- the four partial tiles meet in LDS and are summed in wave order; C/D map: column = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                part[(wave * 32 + f * 16 + (lane >> 4) * 4 + r) * 32 + ct * 16 + (lane & 15)] = acc[f][ct][r];
    __syncthreads();
    if (m < 0) return;
    f32x4 s = *(const f32x4*)(part + row * 32 + c4 * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) s += *(const f32x4*)(part + (w * 32 + row) * 32 + c4 * 4);
    if (p.ksplit > 1) {                                          // K ranges over workgroups: slab for conv_splitk_reduce
        *(f32x4*)(p.ws + ((long)ks * p.M + m) * p.Cout_pad + n) = s;
        return;
    }
    if (n >= p.Cout) return;
    if (p.vec_epi) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = act_apply(fmaf(s[c], wsc[c], bs4[c]), p.act);
        if (p.res0) v += rv0;
        if (p.res1) v += rv1;
        *(f32x4*)(p.dst + po.d + co.d) = v;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (n + c >= p.Cout) break;
            PixOff cc;
            chan_offsets(p, n + c, cc);
            epi_store(p, po, cc, p.bias[n + c], p.wscale ? p.wscale[n + c] : 1.f, s[c]);
        }
    }
}

// dynamic LDS: the staged image (hi + lo) and, behind it, the four partial tiles (which also hold the pixel table first)
inline size_t small_lds_bytes(int nchw, int npix) { return (size_t)2 * nchw * npix * 32 * sizeof(_Float16) + 4 * 32 * 32 * sizeof(float); }

hipError_t launch_small(const SmallK& k, dim3 grid, hipStream_t s, int pk);

}  // namespace fusg
