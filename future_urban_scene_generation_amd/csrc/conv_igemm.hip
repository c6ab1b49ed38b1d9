// Host side of the fused implicit-GEMM convolution: descriptor validation, tile / split-K
// heuristics, launch, and the deterministic split-K slab reduction.  Kernel: conv_kernel.h.
#include <stdlib.h>
#include <mutex>
#include "conv_kernel_tapunit.h"
#include "conv_kernel_small.h"
#include "conv_kernel_tapunit_f32.h"

namespace fusg {

// Deterministic split-K slab reduction + the same epilogue.  One thread per (phase, m, 4 channels).
__global__ __launch_bounds__(256) void conv_splitk_reduce(const ConvK p, int nphase) {
    const int n4 = p.Cout_pad >> 2;
    const long total = (long)nphase * p.M * n4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % n4);
    const long pm = idx / n4;
    const int m = (int)(pm % p.M);
    const int phase = (int)(pm / p.M);
    // slabs are summed in slab order (the result does not depend on the unrolling); eight independent 16-byte loads are
    // in flight per thread instead of one load per dependent add (the loop used to pay one L2 round trip per slab)
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* base = p.ws + ((long)phase * p.ksplit * p.M + m) * p.Cout_pad + c4 * 4;
    const long slab = (long)p.M * p.Cout_pad;
    int k = 0;
    for (; k + 8 <= p.ksplit; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(base + (long)(k + j) * slab);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    if (k + 4 <= p.ksplit) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *(const f32x4*)(base + (long)(k + j) * slab);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += v[j];
        k += 4;
    }
    for (; k < p.ksplit; ++k) s += *(const f32x4*)(base + (long)k * slab);
    PixOff po;
    if (!pix_offsets(p, phase, m, po)) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = c4 * 4 + c;
        if (n < p.Cout) {
            PixOff co;
            chan_offsets(p, n, co);
            epi_store(p, po, co, p.bias[n], p.wscale ? p.wscale[n] : 1.f, s[c]);
        }
    }
}

// Pointwise convolution from at most 8 input channels (the VUnet's NiN stems, 6 -> 128 and 3 -> 32 at full resolution):
// such a layer is a stream of output stores (1.07 GB for 6 -> 128 at B = 32) with 3 - 8 multiply-adds per output
// element - nothing for the matrix cores.  A workgroup takes PW_ITEMS * 256 / n4 consecutive pixels (n4 = cout_pad / 4):
//   1. its pixels' input channels are loaded ONCE (one 16-byte load per thread), pre-processed (ELU / ReLU) and parked in
//      LDS - with one thread per (pixel, 4 output channels) every thread of a pixel would otherwise redo the 8 ELUs;
//   2. a thread's 4 output columns are the same for all of its items (256 % n4 == 0): their 8 weight vectors and the bias
//      are read once into registers;
//   3. per item: the pixel's 8 values from LDS (two broadcast reads), an fp32 fmaf chain per output channel in the exact-
//      fp32 MFMA kernel's k order (bit for bit what that kernel computes: no fp16 split, no range status to raise), bias /
//      activation / residuals, ONE fully coalesced 16-byte store per lane.
// (Measured on the way: 4 columns per thread with per-thread ELUs and 64-bit index divisions 0.47 ms for 6 -> 128 at B = 32,
// 16 columns per thread - four 16-byte stores 64 bytes apart between lanes - 0.62 ms; the tap-unit MFMA kernel 0.31 ms.)
constexpr int PW_ITEMS = 16;
__global__ __launch_bounds__(256) void conv_pointwise_small(const ConvK p, int npix) {
    extern __shared__ __attribute__((aligned(16))) float smem_pw[];
    const int n4 = p.Cout_pad >> 2;                    // threads per pixel; divides 256 (host-checked)
    const int ppb = PW_ITEMS * 256 / n4;               // pixels per workgroup
    float* wl = smem_pw;                               // [8 channels][Cout_pad]
    float* xs = smem_pw + p.Cout_pad * 8;              // [ppb][8]: pre-processed inputs
    const int t = threadIdx.x;
    for (int i = t; i < p.Cout_pad * 8; i += 256) {
        const int c = i / p.Cout_pad, n = i - c * p.Cout_pad;
        wl[i] = c < p.C0 ? p.wpack[(long)n * p.K_pad + c] : 0.f;
    }
    const int hw = p.Ho * p.Wo;
    const int pix0 = blockIdx.x * ppb;
    const int halves = p.C0 > 4 ? 2 : 1;               // 16-byte pieces per pixel
    for (int i = t; i < ppb * 2; i += 256) {
        const int pl = i >> 1, hf = i & 1, m = pix0 + pl;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < npix && hf < halves) {
            const int b = m / hw, rem = m - b * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo;
            v = *(const f32x4*)(p.src0 + ((long)(b * p.H + oy) * p.W + ox) * p.Cs0 + hf * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (p.pre_op == FUSG_PRE_ELU) v[c] = elu1(v[c]);
                else if (p.pre_op == FUSG_PRE_RELU) v[c] = fmaxf(v[c], 0.f);
            }
        }
        *(f32x4*)(xs + pl * 8 + hf * 4) = v;
    }
    __syncthreads();
    const int c4 = t % n4, n = c4 * 4, pl0 = t / n4, ppi = 256 / n4;     // this thread's columns; pixels advance by ppi per item
    if (n >= p.Cout) return;
    f32x4 w[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) w[c] = *(const f32x4*)(wl + c * p.Cout_pad + n);
    const f32x4 bias = *(const f32x4*)(p.bias + n);
#pragma unroll 4
    for (int it = 0; it < PW_ITEMS; ++it) {
        const int pl = it * ppi + pl0, m = pix0 + pl;
        if (m >= npix) return;
        const f32x4 xa = *(const f32x4*)(xs + pl * 8), xb = *(const f32x4*)(xs + pl * 8 + 4);
        // k order 0, 4, 1, 5, 2, 6, 3, 7: v_mfma_f32_32x32x2_f32 takes k = e from lane half 0 and k = 4 + e from half 1 in the
        // exact-fp32 kernel; channels past C0 meet zero weights
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(xb[e], w[4 + e][j], fmaf(xa[e], w[e][j], acc[j]));
        const int b = m / hw, rem = m - b * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const long Y = (long)oy * p.osy + p.ooy[0], X = (long)ox * p.osx + p.oox[0];
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_apply(fmaf(acc[j], 1.f, bias[j]), p.act);
        if (p.res0) v += *(const f32x4*)(p.res0 + b * p.r0n + Y * p.r0h + X * p.r0w + n);
        if (p.res1) v += *(const f32x4*)(p.res1 + b * p.r1n + Y * p.r1h + X * p.r1w + n);
        *(f32x4*)(p.dst + b * p.dsn + Y * p.dsh + X * p.dsw + p.dst_c_off + n) = v;
    }
}

struct TileCfg { int bm, bn; };
static const TileCfg kTiles[] = {{0, 0}, {128, 128}, {128, 64}, {128, 32}, {64, 64}, {64, 128}};

// 256 zero bytes per device, allocated on first use (see ConvK::zeros)
const float* zero_line() {
    static std::mutex mu;
    static float* z[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (!z[dev]) {
        float* q = nullptr;
        if (hipMalloc((void**)&q, 256) != hipSuccess) return nullptr;
        // one-time: the memset runs on the null stream, which non-blocking streams do not wait for
        if (hipMemset(q, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(q); return nullptr; }
        z[dev] = q;
    }
    return z[dev];
}

// Does this launch take the small-spatial kernel (conv_kernel_small.h), and with which K ranges?  Shared by the planner
// (workspace size) and the launcher.
struct SmallCfg { int nimg, rpi, rpi_shift, tpi, wide, RIN, WIN, NPIX, nchw, ksplit, nch32, ntaps; long mt; };
static bool small_cfg(const fusg_conv_desc* d, int precision, SmallCfg* out) {
    // (FUSG_NO_SMALL is read per call: tests and the A/B tools flip it at run time)
    if (getenv("FUSG_NO_SMALL") != nullptr || env_switches().no_halo) return false;
    const int nphase = d->nphase > 0 ? d->nphase : 1;
    if (precision != FUSG_PREC_F16X3 || nphase != 1 || d->kh < 1 || d->kw < 1 || d->kh > 3 || d->kw > 3 || d->dil != 1 ||
        (d->pad_mode != FUSG_PAD_ZERO && d->pad_mode != FUSG_PAD_REPLICATE) || d->upsample != 0 || (d->stride != 1 && d->stride != 2) ||
        d->tile_list || d->stats_out || !d->wfrag || (((uintptr_t)d->wfrag) & 15) != 0 || (d->wfrag_order != 0 && d->wfrag_order != 1))
        return false;
    const int ntaps = d->kh * d->kw;
    if (d->c0k <= 0 || d->c0k % 32 || d->k_pad % ntaps) return false;
    const int ctot = d->k_pad / ntaps, c1k = ctot - d->c0k;
    if (c1k < 0 || c1k % 32 || (c1k > 0 && !d->src1.data)) return false;
    if (d->wfrag_order == 1 && !(d->stride == 2 && d->kh == 3 && d->kw == 3 && d->pad_h == 1 && d->pad_w == 1)) return false;
    const int Ho = d->qh, Wo = d->qw, hw = Ho * Wo;
    // a WINDOW of a larger output (the ring launches of the ICN's up-convolutions: one edge row / column / corner) can use no
    // other fast kernel, whatever its size; whole images only up to small_maxhw pixels (beyond, the halo kernel is faster)
    const long full_h = (d->src0.h + 2L * d->pad_h - d->kh) / d->stride + 1, full_w = (d->src0.w + 2L * d->pad_w - d->kw) / d->stride + 1;
    const bool part = (d->q_oy | d->q_ox) != 0 || Ho != full_h || Wo != full_w;
    const bool window = part && d->pad_mode == FUSG_PAD_REPLICATE;     // (zero padding: whole images only - q_size launches keep their kernels)
    if (part && !window) return false;
    if (!window && hw > env_switches().small_maxhw) return false;
    SmallCfg c;
    c.tpi = 1; c.wide = 0; c.rpi = SMALL_ROWS; c.rpi_shift = 5;
    const long B = d->src0.n;
    if (hw < SMALL_ROWS && SMALL_ROWS % hw == 0 && !window) {          // several whole images per workgroup
        c.nimg = SMALL_ROWS / hw; c.rpi = hw;
        c.rpi_shift = 0;
        while ((1 << c.rpi_shift) < c.rpi) ++c.rpi_shift;
        c.RIN = (Ho - 1) * d->stride + d->kh;
        c.WIN = (Wo - 1) * d->stride + d->kw;
        c.mt = (B + c.nimg - 1) / c.nimg;
    } else {                                                            // runs of 32 pixels of one image (the last run of an image may be short)
        c.nimg = 1;
        c.tpi = (hw + SMALL_ROWS - 1) / SMALL_ROWS;
        if (Wo >= SMALL_ROWS) {
            if (!(Ho == 1 || Wo % SMALL_ROWS == 0)) return false;       // a run must stay inside one output row
            c.wide = 1;
            c.RIN = d->kh;
            c.WIN = (SMALL_ROWS - 1) * d->stride + d->kw;
        } else {
            if (SMALL_ROWS % Wo) return false;
            c.RIN = (SMALL_ROWS / Wo - 1) * d->stride + d->kh;
            c.WIN = (Wo - 1) * d->stride + d->kw;
        }
        c.mt = B * c.tpi;
    }
    c.NPIX = c.nimg * c.RIN * c.WIN;
    if (c.NPIX > SMALL_MAXPIX || c.mt * c.tpi >= (1L << 31) || c.mt * (d->cout_pad / 32) >= (1L << 31) || (long)hw * c.tpi >= (1L << 16)) return false;
    c.nch32 = ctot / 32; c.ntaps = ntaps;
    // chunks per K range: the staged image must fit (<= 112 KiB) and a wave should not need more than two rounds of weights
    int nchw = (112 * 1024) / (c.NPIX * 128);
    const int by_rounds = (2 * 4 * SMALL_NS) / ntaps;
    if (by_rounds < nchw) nchw = by_rounds;
    if (nchw < 1) return false;
    if (d->ksplit == 1 && nchw < c.nch32) return false;            // the caller wants K whole
    if (nchw > c.nch32) nchw = c.nch32;
    c.ksplit = (c.nch32 + nchw - 1) / nchw;
    c.nchw = (c.nch32 + c.ksplit - 1) / c.ksplit;
    c.ksplit = (c.nch32 + c.nchw - 1) / c.nchw;                    // every range non-empty
    // K ranges over workgroups (+ the slab reduce) are implemented and tested, but measured SLOWER than the generic gather
    // they would replace (profiles/r04_ab_experiments.txt: 1024 -> 512 k3 at 4 x 4, B = 32: 75.6 vs 41.6 us - sixteen row
    // tiles each stream the 18.9 MB of weights; 512 -> 128: 20.6 vs 16.5 us): FUSG_SMALL_KSPLIT=1 lets them through
    if (c.ksplit > 1 && getenv("FUSG_SMALL_KSPLIT") == nullptr) return false;
    *out = c;
    return true;
}

static int64_t plan_impl(fusg_conv_desc* d) {
    const long M = (long)d->src0.n * d->qh * d->qw;
    const int nphase = d->nphase > 0 ? d->nphase : 1;
    {
        SmallCfg sc;
        if (small_cfg(d, d->precision == FUSG_PREC_BF16 ? FUSG_PREC_F16X3 : d->precision, &sc)) {
            d->tile = FUSG_TILE_128x32;
            d->ksplit = sc.ksplit;
            return sc.ksplit > 1 ? (int64_t)sc.ksplit * M * d->cout_pad * (int64_t)sizeof(float) : 0;
        }
    }
    if (d->tile == FUSG_TILE_AUTO) {
        int bn = d->cout_pad >= 128 && d->cout_pad % 128 == 0 ? 128 : (d->cout_pad % 64 == 0 ? 64 : 32);
        int bm = 128;
        // small problems: prefer more, smaller tiles
        long tiles = ((M + 127) / 128) * (d->cout_pad / bn) * nphase;
        if (tiles < 256 && bn == 128) { bn = 64; tiles = ((M + 127) / 128) * (d->cout_pad / bn) * nphase; }
        if (tiles < 256 && bn == 64) { bm = 64; }
        d->tile = bm == 128 ? (bn == 128 ? FUSG_TILE_128x128 : bn == 64 ? FUSG_TILE_128x64 : FUSG_TILE_128x32)
                            : FUSG_TILE_64x64;
        // Split-fp16 generic kernel: staging-bound rather than matrix-bound, so the 64-row tiles (32
        // accumulator VGPRs per 64x128, 48 KiB LDS -> 3 workgroups/CU) beat the 128-row ones by 10-20 %
        // on every large layer measured (tools/bench_layer.py); the 32-wide tile only exists as 128x32.
        if (d->precision == FUSG_PREC_F16X3 && bm == 128 && bn >= 64)
            d->tile = bn == 128 ? FUSG_TILE_64x128 : FUSG_TILE_64x64;
    }
    const TileCfg tc = kTiles[d->tile];
    const long tiles = ((M + tc.bm - 1) / tc.bm) * (d->cout_pad / tc.bn) * nphase;
    const int nk = d->k_pad / BK;
    if (d->ksplit <= 0) {
        int ks = 1;
        if (tiles < 192 && nk >= 8) {
            ks = (int)((512 + tiles - 1) / tiles);
            if (ks > nk / 4) ks = nk / 4;
            if (ks > 32) ks = 32;
            if (ks < 1) ks = 1;
        }
        d->ksplit = ks;
    }
    if (d->ksplit <= 1) { d->ksplit = 1; return 0; }
    return (int64_t)nphase * d->ksplit * M * d->cout_pad * (int64_t)sizeof(float);
}

}  // namespace fusg

using namespace fusg;

extern "C" int64_t fusg_conv2d_plan(fusg_conv_desc* d) { return plan_impl(d); }

static int conv2d_impl(const fusg_conv_desc* din, void* stream) {
    fusg_conv_desc dd = *din;
    fusg_conv_desc* d = &dd;
    // FUSG_PREC_BF16: single-pass bf16 on the halo kernel where the layer qualifies, F16X3 everywhere else
    const bool want_bf16 = d->precision == FUSG_PREC_BF16;
    if (want_bf16) d->precision = FUSG_PREC_F16X3;
    hipStream_t s = (hipStream_t)stream;
    const fusg_tensor& x0 = d->src0;
    FUSG_CHECK(is_nhwc(x0), "conv2d: src0 must be NHWC-physical f32 (sc=1, Cs%%4=0, 16B aligned)");
    const bool has1 = d->src1.data != nullptr;
    if (has1) {
        FUSG_CHECK(is_nhwc(d->src1) && same_nhw(x0, d->src1), "conv2d: src1 must be NHWC-physical with src0's n,h,w");
    }
    FUSG_CHECK(d->bias && d->ktab, "conv2d: bias/ktab missing");
    FUSG_CHECK(d->precision >= FUSG_PREC_F32 && d->precision <= FUSG_PREC_EMU_BF16X2, "conv2d: precision %d", din->precision);
    if (d->precision != FUSG_PREC_F16X3) FUSG_CHECK(d->wpack, "conv2d: wpack missing");
    else FUSG_CHECK(d->wpack_h && (((uintptr_t)d->wpack_h) & 15) == 0 && d->wscale && (((uintptr_t)d->wscale) & 15) == 0 && d->status,
                    "conv2d: F16X3 needs 16B-aligned wpack_h and wscale, and a status word");
    FUSG_CHECK((((uintptr_t)d->wpack) & 15) == 0 && (((uintptr_t)d->ktab) & 7) == 0, "conv2d: wpack/ktab misaligned");
    FUSG_CHECK(d->k_pad > 0 && d->k_pad % BK == 0, "conv2d: k_pad %d not a positive multiple of %d", d->k_pad, BK);
    FUSG_CHECK(d->cout > 0 && d->cout_pad % 32 == 0 && d->cout <= d->cout_pad, "conv2d: bad cout %d / cout_pad %d", d->cout, d->cout_pad);
    FUSG_CHECK(d->c0k >= 0 && d->c0k % 4 == 0 && d->c0k <= x0.sw, "conv2d: bad c0k %d (src0 Cs %ld)", d->c0k, (long)x0.sw);
    FUSG_CHECK(d->stride == 1 || d->stride == 2, "conv2d: stride %d", d->stride);
    FUSG_CHECK(d->upsample == 0 || d->upsample == 1, "conv2d: upsample %d", d->upsample);
    FUSG_CHECK(d->pad_mode == FUSG_PAD_ZERO || d->pad_mode == FUSG_PAD_REFLECT || d->pad_mode == FUSG_PAD_REPLICATE, "conv2d: pad_mode");
    FUSG_CHECK(!d->tile_list || (d->tile_count > 0 && !d->stats_out), "conv2d: tile_list needs tile_count > 0 and no stats_out");
    FUSG_CHECK(d->pre_op >= 0 && d->pre_op <= FUSG_PRE_AFFINE, "conv2d: pre_op");
    FUSG_CHECK(d->act >= 0 && d->act <= FUSG_ACT_TANH01, "conv2d: act");
    if (d->pre_op >= FUSG_PRE_AFFINE_RELU) {
        FUSG_CHECK(d->pre_scale && d->pre_shift && ((((uintptr_t)d->pre_scale) | ((uintptr_t)d->pre_shift)) & 15) == 0 &&
                   d->pre_bstride % 4 == 0, "conv2d: affine pre-op needs 16B-aligned scale/shift");
    }
    const int nphase = d->nphase > 0 ? d->nphase : 1;
    FUSG_CHECK(nphase == 1 || nphase == 4, "conv2d: nphase %d", nphase);
    FUSG_CHECK(d->qh > 0 && d->qw > 0, "conv2d: empty output grid");
    FUSG_CHECK(d->q_oy >= 0 && d->q_ox >= 0 && (d->store_mode == FUSG_STORE_NORMAL || (d->q_oy | d->q_ox) == 0) &&
               !(d->stats_out && (d->q_oy | d->q_ox)), "conv2d: q-space origin (%d, %d)", d->q_oy, d->q_ox);
    if (nphase == 1 && d->kh > 0 && d->kw > 0) {
        // the q window must stay inside the convolution's own output range: every tap then lands within one
        // padding width of the (virtual) input, which is all the gathers' single reflection / clamp handles
        const long Hv = x0.h << d->upsample, Wv = x0.w << d->upsample;
        // (zero padding bounds-checks every tap, so any window is safe there)
        FUSG_CHECK(d->dil >= 1 && d->pad_h >= 0 && d->pad_w >= 0 &&
                   (d->pad_mode == FUSG_PAD_ZERO ||
                    ((long)(d->q_oy + d->qh - 1) * d->stride + (long)(d->kh - 1) * d->dil - d->pad_h <= Hv - 1 + d->pad_h &&
                     (long)(d->q_ox + d->qw - 1) * d->stride + (long)(d->kw - 1) * d->dil - d->pad_w <= Wv - 1 + d->pad_w &&
                     d->pad_h <= Hv - 1 && d->pad_w <= Wv - 1)),
                   "conv2d: q window [%d+%d, %d+%d] reaches outside the padded input", d->q_oy, d->qh, d->q_ox, d->qw);
    }
    const long Ml = (long)x0.n * d->qh * d->qw;
    FUSG_CHECK(Ml > 0 && Ml < (1L << 31), "conv2d: M out of range");
    FUSG_CHECK(x0.n * x0.h * x0.w * x0.sw < (1L << 40), "conv2d: src too large");
    const fusg_tensor& o = d->dst;
    FUSG_CHECK(o.data && o.dtype == FUSG_F32 && o.n == x0.n, "conv2d: dst missing / batch mismatch");
    // logical output extents implied by the store mode
    long oh, ow, oc;
    if (d->store_mode == FUSG_STORE_D2S) {
        FUSG_CHECK(d->cout % 4 == 0 && nphase == 1, "conv2d: D2S needs cout%%4==0");
        oh = 2L * d->qh; ow = 2L * d->qw; oc = d->cout / 4;
    } else if (d->store_mode == FUSG_STORE_S2D) {
        FUSG_CHECK(d->qh % 2 == 0 && d->qw % 2 == 0 && nphase == 1, "conv2d: S2D needs even output grid");
        oh = d->qh / 2; ow = d->qw / 2; oc = 4L * d->cout;
    } else {
        FUSG_CHECK(d->store_mode == FUSG_STORE_NORMAL && d->out_sy >= 1 && d->out_sx >= 1, "conv2d: store mode / out stride");
        oh = 0; ow = 0;
        for (int ph = 0; ph < nphase; ++ph) {
            FUSG_CHECK(d->out_oy[ph] >= 0 && d->out_ox[ph] >= 0, "conv2d: negative phase offset");
            const long eh = (long)(d->q_oy + d->qh - 1) * d->out_sy + d->out_oy[ph] + 1, ew = (long)(d->q_ox + d->qw - 1) * d->out_sx + d->out_ox[ph] + 1;
            oh = oh > eh ? oh : eh;
            ow = ow > ew ? ow : ew;
        }
        oc = d->cout;
    }
    FUSG_CHECK(oh <= o.h && ow <= o.w && d->dst_c_off >= 0 && d->dst_c_off + oc <= o.c,
               "conv2d: dst [%ld,%ld,%ld,%ld] too small for output %ldx%ldx%ld at channel offset %d",
               (long)o.n, (long)o.c, (long)o.h, (long)o.w, oc, oh, ow, d->dst_c_off);
    const fusg_tensor* rs[2] = {&d->res0, &d->res1};
    for (int i = 0; i < 2; ++i) {
        if (!rs[i]->data) continue;
        FUSG_CHECK(d->store_mode == FUSG_STORE_NORMAL && rs[i]->dtype == FUSG_F32 && rs[i]->n == o.n &&
                   rs[i]->c >= oc && rs[i]->h >= oh && rs[i]->w >= ow, "conv2d: residual %d shape mismatch", i);
    }

    plan_impl(d);
    FUSG_CHECK(d->tile >= 1 && d->tile <= 5, "conv2d: tile %d", d->tile);
    const TileCfg tc = kTiles[d->tile];
    FUSG_CHECK(d->cout_pad % tc.bn == 0, "conv2d: cout_pad %d not a multiple of tile N %d", d->cout_pad, tc.bn);
    if (d->ksplit > 1) FUSG_CHECK(d->workspace && (((uintptr_t)d->workspace) & 15) == 0, "conv2d: split-K needs a 16B-aligned workspace");
    if (d->ksplit > 1 && d->splitk_counters)
        FUSG_CHECK((long)((Ml + tc.bm - 1) / tc.bm) * (d->cout_pad / tc.bn) * nphase <= d->splitk_counters_len,
                   "conv2d: %d split-K counters for %ld tiles", d->splitk_counters_len, (long)((Ml + tc.bm - 1) / tc.bm) * (d->cout_pad / tc.bn) * nphase);

    ConvK k;
    memset(&k, 0, sizeof(k));
    k.src0 = (const float*)x0.data; k.src1 = has1 ? (const float*)d->src1.data : (const float*)x0.data;
    k.wpack = d->wpack; k.bias = d->bias; k.ktab = (const int2*)d->ktab;
    k.pre_scale = d->pre_scale; k.pre_shift = d->pre_shift; k.pre_bstride = d->pre_bstride;
    k.dst = (float*)o.data; k.dsn = o.sn; k.dsc = o.sc; k.dsh = o.sh; k.dsw = o.sw;
    if (d->res0.data) { k.res0 = (const float*)d->res0.data; k.r0n = d->res0.sn; k.r0c = d->res0.sc; k.r0h = d->res0.sh; k.r0w = d->res0.sw; }
    if (d->res1.data) { k.res1 = (const float*)d->res1.data; k.r1n = d->res1.sn; k.r1c = d->res1.sc; k.r1h = d->res1.sh; k.r1w = d->res1.sw; }
    k.ws = d->workspace;
    k.counters = d->ksplit > 1 ? d->splitk_counters : nullptr;
    k.qy0 = d->q_oy; k.qx0 = d->q_ox;
    k.zeros = zero_line();
    if (!k.zeros) { set_error("conv2d: cannot allocate the zero line"); return FUSG_ERR_LAUNCH; }
    k.touch_w = env_switches().no_touch ? 0 : 1;
    k.wpack_h = (const _Float16*)d->wpack_h;
    if (d->precision == FUSG_PREC_F16X3) { k.wscale = d->wscale; k.status = d->status; }
    k.round_bits = d->precision == FUSG_PREC_EMU_BF16 ? 8 : (d->precision == FUSG_PREC_EMU_BF16X2 ? 16 : 0);
    k.H = (int)x0.h; k.W = (int)x0.w; k.ups = d->upsample; k.Hv = k.H << k.ups; k.Wv = k.W << k.ups;
    k.Cs0 = (int)x0.sw; k.Cs1 = has1 ? (int)d->src1.sw : (int)x0.sw; k.C0 = d->c0k;
    k.K_pad = d->k_pad; k.nk = d->k_pad / BK; k.Cout = d->cout; k.Cout_pad = d->cout_pad;
    k.stride = d->stride; k.pad_mode = d->pad_mode; k.pre_op = d->pre_op; k.act = d->act; k.store_mode = d->store_mode;
    k.B = (int)x0.n; k.Ho = d->qh; k.Wo = d->qw; k.M = (int)Ml;
    k.MT = (k.M + tc.bm - 1) / tc.bm; k.NT = d->cout_pad / tc.bn;
    k.osy = d->out_sy; k.osx = d->out_sx;
    for (int i = 0; i < 4; ++i) { k.ooy[i] = d->out_oy[i]; k.oox[i] = d->out_ox[i]; }
    k.dst_c_off = d->dst_c_off; k.Cd = d->store_mode == FUSG_STORE_D2S ? d->cout / 4 : d->cout;
    k.ksplit = d->ksplit; k.steps_per_split = (k.nk + d->ksplit - 1) / d->ksplit;
    {   // 16-byte epilogue accesses need channel-contiguous, 16-byte aligned destinations and residuals
        auto vec_ok = [](const fusg_tensor& t) {
            return t.sc == 1 && (((uintptr_t)t.data) & 15) == 0 && t.sw % 4 == 0 && t.sh % 4 == 0 && t.sn % 4 == 0;
        };
        bool v = d->cout % 4 == 0 && d->dst_c_off % 4 == 0 && vec_ok(o) && !env_switches().no_vec_epi;
        if (d->store_mode == FUSG_STORE_D2S) v = v && (d->cout / 4) % 4 == 0;
        if (d->res0.data) v = v && vec_ok(d->res0);
        if (d->res1.data) v = v && vec_ok(d->res1);
        k.vec_epi = v ? 1 : 0;
    }
    if (d->stats_out) {
        if (!(k.vec_epi && d->act == FUSG_ACT_NONE && !d->res0.data && !d->res1.data && d->store_mode == FUSG_STORE_NORMAL &&
              nphase == 1 && d->ksplit <= 1 && ((long)d->qh * d->qw) % 32 == 0)) {
            set_error("conv2d: fused statistics need act NONE, no residual, NORMAL store, nphase 1, ksplit 1, qh*qw%%32==0 "
                      "and a channel-contiguous aligned dst with cout%%4==0");
            return FUSG_ERR_UNSUPPORTED;
        }
        k.stats = d->stats_out;
        k.stats_slots = (int)(((long)d->qh * d->qw) / 32);
        if (d->stats_slots > 0) {
            if (d->stats_slots < k.stats_slots) { set_error("conv2d: stats_slots %d < %d slots of this launch", d->stats_slots, k.stats_slots); return FUSG_ERR_INVALID; }
            k.stats_slots = d->stats_slots;
        }
    }

    dim3 grid(k.MT * k.NT, nphase, d->ksplit);
    const double flops = 2.0 * (double)Ml * d->cout * d->k_pad * nphase;   // padded-K flops; bench uses algorithmic ones
    prof_begin(0, s, flops);
    hipError_t e;
    int pk = PK_NONE;
    k.pre_relu = 0;
    switch (d->pre_op) {
        case FUSG_PRE_RELU: k.pre_relu = 1; break;
        case FUSG_PRE_ELU: pk = PK_ELU; break;
        case FUSG_PRE_AFFINE_RELU: pk = PK_AFFINE; k.pre_relu = 1; break;
        case FUSG_PRE_AFFINE: pk = PK_AFFINE; break;
        default: break;
    }
    const bool gen = d->pad_mode != FUSG_PAD_ZERO || d->upsample != 0;
    // halo kernel: stride 1 (any dilation / padding mode / fused upsample), or stride 2 in parity-quadrant form
    // (wfrag_order 1: k3/k4, pad 1, one source, even H and W); everything else takes the generic gather
    const bool s2d_form = d->wfrag_order == 1 && d->stride == 2 && x0.h % 2 == 0 && x0.w % 2 == 0;
    const bool f32_halo = d->precision == FUSG_PREC_F32 && d->wfrag_f32 != nullptr && (((uintptr_t)d->wfrag_f32) & 15) == 0 &&
                          getenv("FUSG_NO_F32_HALO") == nullptr;
    const bool halo_ok = (d->precision == FUSG_PREC_F16X3 || f32_halo) && nphase == 1 && d->ksplit <= 1 &&
                         ((d->stride == 1 && d->wfrag_order == 0) || (s2d_form && d->upsample == 0)) &&
                         d->kh >= 1 && d->kw >= 1 && d->dil >= 1 && d->c0k % 32 == 0 && d->c0k > 0 &&
                         (!has1 || (d->k_pad / (d->kh * d->kw) - d->c0k) % 32 == 0) && d->qh % 8 == 0 && d->qw % 16 == 0 &&
                         d->k_pad % (d->kh * d->kw) == 0 && d->wfrag != nullptr && (((uintptr_t)d->wfrag) & 15) == 0 &&
                         !env_switches().no_halo && (d->q_oy | d->q_ox) == 0;
    // small images (<= 16 x 16): the latency-built kernel of conv_kernel_small.h
    {
        SmallCfg sc;
        fusg_conv_desc probe = *din;                    // (plan_impl above has already overwritten d->ksplit / d->tile)
        if (small_cfg(&probe, d->precision, &sc) && sc.ksplit == d->ksplit) {
            SmallK h;
            memset(&h, 0, sizeof(h));
            h.c = k;
            h.kh = d->kh; h.kw = d->kw; h.pad_h = d->pad_h; h.pad_w = d->pad_w; h.stride = d->stride;
            h.nch0 = d->c0k / 32; h.nch32 = sc.nch32; h.ntaps = sc.ntaps;
            h.wfrag = (const _Float16*)d->wfrag; h.nt32 = d->cout_pad / 32;
            if (d->wfrag_order == 1) {                  // parity-quadrant slab order of the stride-2 3x3 layers (pack.s2d_tap_order)
                int slab = 0;
                for (int q = 0; q < 4; ++q)
                    for (int ky = 0; ky < 3; ++ky) {
                        if (((ky - 1) & 1) != (q >> 1)) continue;
                        for (int kx = 0; kx < 3; ++kx)
                            if (((kx - 1) & 1) == (q & 1)) h.tapslab |= (unsigned long long)(slab++) << (4 * (ky * 3 + kx));
                    }
            } else {
                for (int tp = 0; tp < sc.ntaps; ++tp) h.tapslab |= (unsigned long long)tp << (4 * tp);
            }
            h.nchw = sc.nchw; h.nimg = sc.nimg; h.rpi = sc.rpi; h.rpi_shift = sc.rpi_shift; h.tpi = sc.tpi; h.wide = sc.wide;
            h.pad_mode = d->pad_mode;
            h.RIN = sc.RIN; h.WIN = sc.WIN; h.NPIX = sc.NPIX;
            auto magic = [](int dv) -> unsigned { return dv < 2 ? 0u : (unsigned)(((1UL << 32) + (unsigned long)dv - 1) / (unsigned long)dv); };
            h.m_wo = magic(d->qw); h.m_hw = magic(d->qh * d->qw); h.m_win = magic(sc.WIN); h.m_rw = magic(sc.RIN * sc.WIN);
            h.m_npix = magic(sc.NPIX); h.m_taps = magic(sc.ntaps); h.m_tpi = magic(sc.tpi);
            // (the reciprocals are exact while dividend * divisor < 2^32: the dividends are tile indices (< 2^31 / tpi, checked in
            // small_cfg), pixel indices inside a 32-row tile and step numbers - the planner and this launcher must agree, or a
            // layer planned for this kernel would run on the generic gather with this kernel's tile / split choice)
            {
                h.c.MT = (int)sc.mt; h.c.NT = d->cout_pad / 32;
                h.c.ksplit = sc.ksplit;
                dim3 sgrid(h.c.MT * h.c.NT, 1, sc.ksplit);
                e = launch_small(h, sgrid, s, pk);
                if (e != hipSuccess) { set_error("conv2d small-image launch: %s", hipGetErrorString(e)); prof_end(0, s); return FUSG_ERR_LAUNCH; }
                if (sc.ksplit > 1) {
                    const long total = (long)k.M * (k.Cout_pad >> 2);
                    ConvK kr = h.c;
                    hipLaunchKernelGGL(conv_splitk_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, kr, 1);
                    e = hipGetLastError();
                    if (e != hipSuccess) { set_error("conv2d split-K reduce launch: %s", hipGetErrorString(e)); prof_end(0, s); return FUSG_ERR_LAUNCH; }
                }
                note_conv_kernel(FUSG_CONV_SMALL);
                prof_end(0, s);
                return FUSG_OK;
            }
        }
    }
    // pointwise from <= 8 channels: the streaming VALU kernel above - an exact fp32 fmaf chain in the generic fp32 kernel's k
    // order, so it serves both the split-fp16 and (round 4) the exact-fp32 precision with that kernel's bits
    // (FUSG_NO_POINTWISE, read per call, keeps the MFMA kernels: tests compare the two)
    if ((d->precision == FUSG_PREC_F16X3 || d->precision == FUSG_PREC_F32) && getenv("FUSG_NO_POINTWISE") == nullptr && nphase == 1 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->upsample == 0 && !has1 &&
        d->c0k >= 4 && d->c0k <= 8 && d->ksplit <= 1 && d->store_mode == FUSG_STORE_NORMAL && k.vec_epi && !d->stats_out &&
        d->cout % 4 == 0 && d->pre_op <= FUSG_PRE_ELU && (d->q_oy | d->q_ox) == 0 && !d->tile_list && d->wpack && d->pad_h == 0 &&
        d->pad_w == 0 && 256 % (d->cout_pad >> 2) == 0 && !env_switches().no_halo && !env_switches().no_pointwise) {
        const int n4 = d->cout_pad >> 2, ppb = PW_ITEMS * 256 / n4;
        hipLaunchKernelGGL(conv_pointwise_small, dim3((unsigned)((Ml + ppb - 1) / ppb)), dim3(256),
                           (size_t)(d->cout_pad * 8 + ppb * 8) * sizeof(float), s, k, (int)Ml);
        e = hipGetLastError();
        prof_end(0, s);
        if (e != hipSuccess) { set_error("conv2d pointwise launch: %s", hipGetErrorString(e)); return FUSG_ERR_LAUNCH; }
        note_conv_kernel(FUSG_CONV_POINTWISE);
        return FUSG_OK;
    }
    // few-channel k x k layers (the 7x7 stems) in exact fp32 (round 4): conv_kernel_tapunit_f32.h
    if (d->wfrag_order == 2 && d->precision == FUSG_PREC_F32 && d->wfrag_f32 != nullptr && (((uintptr_t)d->wfrag_f32) & 15) == 0 &&
        getenv("FUSG_NO_F32_HALO") == nullptr) {
        const int nunits = d->kh * d->kw * (d->c0k / 4);
        const bool ok = nphase == 1 && d->upsample == 0 && d->ksplit <= 1 && !has1 && d->c0k >= 4 && d->c0k <= 24 && d->c0k % 4 == 0 &&
                        d->kh >= 1 && d->kw >= 1 && d->dil == 1 && d->qh % 8 == 0 && d->qw % 16 == 0 && nunits <= 320 &&
                        d->k_pad >= d->kh * d->kw * d->c0k && (d->q_oy | d->q_ox) == 0 && !d->tile_list && !env_switches().no_halo;
        if (ok) {
            TapUnitF h;
            memset(&h, 0, sizeof(h));
            h.c = k;
            h.stride = d->stride; h.pad_h = d->pad_h; h.pad_w = d->pad_w;
            h.HH = 7 * d->stride + d->kh; h.HW = 15 * d->stride + d->kw;
            h.CP = d->c0k; h.PP = d->c0k;
            h.RP = h.HW * h.PP;
            h.nunits = nunits;
            h.wfrag = (const float*)d->wfrag_f32;
            h.nt32 = d->cout_pad / 32;
            const int upp = d->c0k / 4;
            for (int j = 0; j < nunits; ++j) {
                const int tap = j / upp, u = j - tap * upp, ky = tap / d->kw, kx = tap - ky * d->kw;
                h.uoff[j] = ky * h.RP + kx * h.PP + u * 4;
            }
            int bn = d->cout_pad % 128 == 0 ? 128 : (d->cout_pad % 64 == 0 ? 64 : 32);
            if (const int v = env_switches().halo_bn; (v == 32 || v == 64 || v == 128) && d->cout_pad % v == 0) bn = v;
            h.tiles_x = d->qw / 16; h.tiles_per_img = (d->qh / 8) * h.tiles_x;
            h.c.MT = (int)x0.n * h.tiles_per_img; h.c.NT = d->cout_pad / bn;
            h.c.ksplit = 1; h.c.wscale = nullptr; h.c.status = nullptr;
            if ((size_t)h.HH * h.RP * sizeof(float) <= 80 * 1024 && h.HH * h.HW * (h.CP / 4) <= 256 * 8) {
                dim3 hgrid(h.c.MT * h.c.NT, 1, 1);
                e = bn == 128 ? launch_tapunit_f32_128(h, hgrid, s, pk) : bn == 64 ? launch_tapunit_f32_64(h, hgrid, s, pk)
                                                                                   : launch_tapunit_f32_32(h, hgrid, s, pk);
                if (e != hipSuccess) { set_error("conv2d fp32 tap-unit launch: %s", hipGetErrorString(e)); prof_end(0, s); return FUSG_ERR_LAUNCH; }
                note_conv_kernel(FUSG_CONV_TAPUNIT_F32);
                prof_end(0, s);
                return FUSG_OK;
            }
        }
    }
    // few-channel k x k layers (the 7x7 stems): tap-unit kernel (conv_kernel_tapunit.h)
    if (d->wfrag_order == 2) {
        const int unit = d->c0k % 8 == 0 ? 8 : 4;
        const int nunits = d->kh * d->kw * (d->c0k / unit);
        const bool ok = d->precision == FUSG_PREC_F16X3 && nphase == 1 && d->upsample == 0 && d->ksplit <= 1 && !has1 &&
                        d->wfrag != nullptr && (((uintptr_t)d->wfrag) & 15) == 0 && d->c0k >= 4 && d->c0k <= 24 && d->kh >= 1 &&
                        d->kw >= 1 && d->dil == 1 && d->qh % 8 == 0 && d->qw % 16 == 0 && nunits <= 160 &&
                        d->k_pad >= d->kh * d->kw * d->c0k && (d->q_oy | d->q_ox) == 0 && !d->tile_list &&
                        !env_switches().no_halo;
        if (ok) {
            TapUnitK h;
            memset(&h, 0, sizeof(h));
            h.c = k;
            h.stride = d->stride; h.pad_h = d->pad_h; h.pad_w = d->pad_w;
            h.HH = 7 * d->stride + d->kh; h.HW = 15 * d->stride + d->kw;
            h.CP = d->c0k; h.PP = d->c0k;
            h.RP = (h.HW * h.PP + 127) / 128 * 128;               // rows 256 B apart: conflict-free 16-lane read groups
            h.nunits = nunits; h.nsteps = (nunits + 16 / unit - 1) / (16 / unit);
            h.wfrag = (const _Float16*)d->wfrag;
            h.nt32 = d->cout_pad / 32;
            // FUSG_PREC_BF16: the stem in single-pass bf16 too (wfrag_bf16 in the tap-unit form, pack.py: frag_tapunit_bf16)
            const bool tbf = want_bf16 && d->wfrag_bf16 != nullptr && (((uintptr_t)d->wfrag_bf16) & 15) == 0 && getenv("FUSG_NO_BF16_TAPUNIT") == nullptr;
            if (tbf) { h.wfrag = (const _Float16*)d->wfrag_bf16; h.c.wscale = nullptr; h.c.status = nullptr; }
            const int upp = d->c0k / unit;
            for (int j = 0; j < nunits; ++j) {
                const int tap = j / upp, u = j - tap * upp, ky = tap / d->kw, kx = tap - ky * d->kw;
                h.uoff[j] = ky * h.RP + kx * h.PP + u * unit;
            }
            int bn = d->cout_pad % 128 == 0 ? 128 : (d->cout_pad % 64 == 0 ? 64 : 32);
            // thin layers (a handful of k-steps: the launch is a stream of output stores) run 24-45 % faster on the
            // 64-column tile: its 16 KiB epilogue detour leaves room for more resident workgroups than the 128-column one
            if (bn == 128 && h.nsteps <= 4) bn = 64;
            if (const int v = env_switches().halo_bn; (v == 32 || v == 64 || v == 128) && d->cout_pad % v == 0) bn = v;
            h.tiles_x = d->qw / 16; h.tiles_per_img = (d->qh / 8) * h.tiles_x;
            h.c.MT = (int)x0.n * h.tiles_per_img; h.c.NT = d->cout_pad / bn;
            h.c.ksplit = 1;
            if (tapunit_lds_bytes(h.HH, h.RP) <= 80 * 1024 && h.HH * h.HW * (h.CP / 4) <= 256 * 8) {
                dim3 hgrid(h.c.MT * h.c.NT, 1, 1);
                e = bn == 128 ? launch_tapunit_128(h, hgrid, s, pk, unit, tbf) : bn == 64 ? launch_tapunit_64(h, hgrid, s, pk, unit, tbf)
                                                                                          : launch_tapunit_32(h, hgrid, s, pk, unit, tbf);
                if (e != hipSuccess) { set_error("conv2d tap-unit launch: %s", hipGetErrorString(e)); prof_end(0, s); return FUSG_ERR_LAUNCH; }
                note_conv_kernel(tbf ? FUSG_CONV_TAPUNIT_BF16 : FUSG_CONV_TAPUNIT);
                prof_end(0, s);
                return FUSG_OK;
            }
        }
    }
    if (halo_ok) {
        HaloK h;
        memset(&h, 0, sizeof(h));
        h.c = k;
        h.kh = d->kh; h.kw = d->kw; h.dil = d->dil; h.pad_h = d->pad_h; h.pad_w = d->pad_w;
        h.c1k = d->k_pad / (d->kh * d->kw) - d->c0k;
        h.wfrag = (const _Float16*)d->wfrag;
        h.nt32 = d->cout_pad / 32;
        int bn = d->cout_pad % 128 == 0 ? 128 : (d->cout_pad % 64 == 0 ? 64 : 32);
        // pointwise layers are bound by their output stores, not by operand staging: the 64-column tile (a quarter of
        // the epilogue LDS, more resident workgroups) is 6-24 % faster there (hourglass 1x1s); k x k layers keep 128
        if (bn == 128 && d->kh * d->kw == 1) bn = 64;
        {   // small grids: narrower column tiles until the launch has enough workgroups to fill the chip
            const long min_wg = env_switches().halo_minwg;
            const long mt = (long)x0.n * (d->tile_list ? d->tile_count : (d->qh / 8) * (d->qw / 16));
            while (bn > 32 && mt * (d->cout_pad / bn) < min_wg) bn /= 2;
        }
        if (const int v = env_switches().halo_bn; (v == 32 || v == 64 || v == 128) && d->cout_pad % v == 0) bn = v;
        h.c.NT = d->cout_pad / bn;
        h.c.ksplit = 1;
        h.HH = 7 + (d->kh - 1) * d->dil + 1; h.HW = 15 + (d->kw - 1) * d->dil + 1;
        if (s2d_form) {
            // parity-quadrant form (include/fusg.h, wfrag_order): input row 2Y - 1 + ky = parity (ky-1)&1, sub-row Y + (ky-1 >> 1)
            if (!(d->kh == d->kw && (d->kh == 3 || d->kh == 4) && d->pad_h == 1 && d->pad_w == 1 && d->dil == 1 && !has1 &&
                  d->pad_mode != FUSG_PAD_REPLICATE)) {
                set_error("conv2d: wfrag_order 1 needs stride 2, k3/k4, pad 1, dil 1, one source");
                prof_end(0, s);
                return FUSG_ERR_INVALID;
            }
            h.s2d = 1; h.HH = 10; h.HW = 18;
            int slab = 0;
            for (int q = 0; q < 4; ++q) {
                h.qwoff[q] = slab;
                int n = 0;
                for (int ky = 0; ky < d->kh; ++ky) {
                    if (((ky - 1) & 1) != (q >> 1)) continue;
                    for (int kx = 0; kx < d->kw; ++kx) {
                        if (((kx - 1) & 1) != (q & 1)) continue;
                        const int dy = (ky - 1) >> 1, dx = (kx - 1) >> 1;          // arithmetic shift: floor
                        h.qtdy[q][n] = dy + 1;
                        h.qtdx[q][n] = dx + 1;
                        ++n;
                    }
                }
                h.qtaps[q] = n;
                slab += n;
            }
        }
        // (Round 3 tried 16 x 16 pixel patches for the 32 / 64-column tiles - half the workgroups, twice the work each,
        // halo 1.27x instead of 1.41x of the patch: every such launch got 13 - 51 % SLOWER (32 -> 32 1x1 at 256 x 256
        // 130 -> 186 us, 64 -> 32 3x3 311 -> 351 us), conv time of the pass 23.9 -> 24.5 ms.  Dropped.)
        h.tiles_x = d->qw / 16; h.tiles_per_img = (d->qh / 8) * h.tiles_x;
        h.c.MT = (int)x0.n * h.tiles_per_img;
        if (d->tile_list) {
            if (d->tile_count > h.tiles_per_img) { set_error("conv2d: tile_count %d > %d patches per image", d->tile_count, h.tiles_per_img); prof_end(0, s); return FUSG_ERR_INVALID; }
            h.tile_list = d->tile_list; h.tile_count = d->tile_count;
            h.c.MT = (int)x0.n * d->tile_count;
        }
        if (halo_fits(h.HH, h.HW)) {
            dim3 hgrid(h.c.MT * h.c.NT, 1, 1);
            const bool bf = want_bf16 && d->wfrag_bf16 != nullptr && (((uintptr_t)d->wfrag_bf16) & 15) == 0;
            if (bf) { h.wfrag = (const _Float16*)d->wfrag_bf16; h.c.wscale = nullptr; h.c.status = nullptr; }
            if (f32_halo) { h.wfrag = (const _Float16*)d->wfrag_f32; h.c.wscale = nullptr; h.c.status = nullptr; }
            const int mode = f32_halo ? 2 : (bf ? 1 : 0);
            // bf16 mode on a 16 x 16 pixel patch (conv_kernel_halo.h, BM == 256; FUSG_BF16_BIG=14|22 picks the wave layout).  OFF by
            // default: measured 1.4x (1 x 4 waves) and 3x (2 x 2) SLOWER than the 8 x 16 patch (profiles/r04_ab_experiments.txt) - 128
            // accumulator registers per wave leave one wave per SIMD, and nothing then overlaps the fp32 -> bf16 staging
            if (bf && bn == 128 && !s2d_form && !d->tile_list && d->qh % 16 == 0 && d->qw % 16 == 0) {
                static const int big = [] { const char* v = getenv("FUSG_BF16_BIG"); return v ? atoi(v) : 0; }();
                const int HHb = 15 + (d->kh - 1) * d->dil + 1;
                if (big != 0 && HHb * h.HW * 8 <= 256 * 12) {
                    HaloK hb = h;
                    hb.HH = HHb;
                    hb.tiles_per_img = (d->qh / 16) * hb.tiles_x;
                    hb.c.MT = (int)x0.n * hb.tiles_per_img;
                    dim3 bgrid(hb.c.MT * hb.c.NT, 1, 1);
                    e = big == 22 ? launch_halo_big22(hb, bgrid, s, pk) : launch_halo_big14(hb, bgrid, s, pk);
                    if (e == hipSuccess) {
                        note_conv_kernel(FUSG_CONV_HALO_BF16);
                        prof_end(0, s);
                        return FUSG_OK;
                    }
                    (void)hipGetLastError();                     // does not fit: the 8 x 16 patch below
                }
            }
            // narrow column tiles of k x k layers on SMALL grids: K split over the waves (conv_kernel_halo.h, KS) when every wave
            // gets a tap.  Measured per dispatch inside the pass (round 3, same card): grids of 256 - 1024 workgroups 2 - 14 %
            // shorter (their time is one workgroup's latency); grids of 4096 - 16384 workgroups 19 - 42 % LONGER (the four
            // partial tiles need 64 KiB of LDS: two workgroups per CU instead of three, and those launches move 3 - 4.4 TB/s -
            // they live on bytes in flight).  Hence the limit.
            const int ntaps = d->kh * d->kw;
            const bool ks_ok = !s2d_form && !d->tile_list && !env_switches().no_ksplit && (long)h.c.MT * h.c.NT <= 1024;
            const bool k4 = ks_ok && bn == 32 && ntaps >= 4, k2 = ks_ok && bn == 64 && ntaps >= 2;
            e = bn == 128 ? launch_halo_128(h, hgrid, s, pk, mode)
                          : bn == 64 ? (k2 ? launch_halo_64k(h, hgrid, s, pk, mode) : launch_halo_64(h, hgrid, s, pk, mode))
                                     : (k4 ? launch_halo_32k(h, hgrid, s, pk, mode) : launch_halo_32(h, hgrid, s, pk, mode));
            if (e != hipSuccess) { set_error("conv2d halo launch: %s", hipGetErrorString(e)); prof_end(0, s); return FUSG_ERR_LAUNCH; }
            note_conv_kernel(f32_halo ? FUSG_CONV_HALO_F32 : bf ? FUSG_CONV_HALO_BF16 : (h.s2d ? FUSG_CONV_HALO_S2D : FUSG_CONV_HALO));
            prof_end(0, s);
            return FUSG_OK;
        }
    }
    if (d->tile_list) {
        set_error("conv2d: tile_list needs a launch that qualifies for the halo kernel");
        prof_end(0, s);
        return FUSG_ERR_UNSUPPORTED;
    }
    note_conv_kernel(d->precision == FUSG_PREC_F16X3 ? FUSG_CONV_GENERIC_F16X3 : FUSG_CONV_GENERIC_F32);
    if (d->precision == FUSG_PREC_F16X3) {
        switch (d->tile) {
            case FUSG_TILE_128x128: e = launch_h3_128x128(k, grid, s, pk, gen); break;
            case FUSG_TILE_128x64:  e = launch_h3_128x64(k, grid, s, pk, gen); break;
            case FUSG_TILE_128x32:  e = launch_h3_128x32(k, grid, s, pk, gen); break;
            case FUSG_TILE_64x64:   e = launch_h3_64x64(k, grid, s, pk, gen); break;
            default:                e = launch_h3_64x128(k, grid, s, pk, gen); break;
        }
    } else
    switch (d->tile) {
        case FUSG_TILE_128x128: e = launch_tile_128x128(k, grid, s, pk, gen); break;
        case FUSG_TILE_128x64:  e = launch_tile_128x64(k, grid, s, pk, gen); break;
        case FUSG_TILE_128x32:  e = launch_tile_128x32(k, grid, s, pk, gen); break;
        case FUSG_TILE_64x64:   e = launch_tile_64x64(k, grid, s, pk, gen); break;
        default:                e = launch_tile_64x128(k, grid, s, pk, gen); break;
    }
    if (e != hipSuccess) { set_error("conv2d launch: %s", hipGetErrorString(e)); prof_end(0, s); return FUSG_ERR_LAUNCH; }
    if (d->ksplit > 1 && k.counters == nullptr) {
        const long total = (long)nphase * k.M * (k.Cout_pad >> 2);
        hipLaunchKernelGGL(conv_splitk_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k, nphase);
        e = hipGetLastError();
        if (e != hipSuccess) { set_error("conv2d split-K reduce launch: %s", hipGetErrorString(e)); prof_end(0, s); return FUSG_ERR_LAUNCH; }
    }
    prof_end(0, s);
    return FUSG_OK;
}
extern "C" int fusg_conv2d(const fusg_conv_desc* din, void* stream) { return fusg::plan_dispatch(conv2d_impl, stream, din); }
