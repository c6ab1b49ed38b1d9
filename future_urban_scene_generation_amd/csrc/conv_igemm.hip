// Fused implicit-GEMM convolution for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix
// cores (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, 157 TFLOP/s dense peak).
//
// GEMM view: out[m, n] = sum_k A[m, k] * W[n, k];  m = (b, qy, qx) output pixels, n = output
// channel, k = (tap, concat channel).  A is never materialised: a 256-thread workgroup gathers a
// BM x 32 slice of it straight from the NHWC activations through the per-layer k-table (one int2
// per 4 consecutive k: tap offset + channel offset + source select), applies the fused pre-op
// (ReLU / ELU / per-(b,c) affine+ReLU = eval BatchNorm, InstanceNorm or LayerNorm normalise-on-
// load), handles zero / reflect padding and the optional 2x nearest upsample in the address
// computation, and stages it through LDS ([row][36] floats: conflict-free ds_read_b128 for the
// MFMA operand fetch, conflict-free ds_write_b128 for the staging store).  The weight tile
// ([cout][k], k contiguous, packed once at load time) is staged the same way.  Global loads for
// step s+1 are issued before the MFMAs of step s (register double-buffering + two LDS buffers,
// one barrier per K-step).  The epilogue adds the bias, applies the activation, adds up to two
// residuals and stores through arbitrary destination strides with an optional DepthToSpace /
// SpaceToDepth / transposed-convolution-phase coordinate mapping.  Small-M layers use split-K
// with a deterministic slab reduction.
//
// Wave tiling: 4 waves as WM x WN, each wave owns TM x TN tiles of 32x32 (16 accumulator VGPRs
// each).  MFMA operand maps (cdna_hip_programming.md §3): A lane l holds A[row l&31][k l>>5],
// B lane l holds B[k l>>5][col l&31]; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
#include "common.h"

namespace fusg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;    // k per pipeline step
constexpr int LDK = 36;   // LDS row pitch in floats (144 B): b128 reads and writes conflict-free

struct ConvK {
    const float* src0; const float* src1;
    const float* wpack; const float* bias; const int2* ktab;
    const float* pre_scale; const float* pre_shift;
    float* dst; const float* res0; const float* res1; float* ws;
    long dsn, dsc, dsh, dsw;
    long r0n, r0c, r0h, r0w;
    long r1n, r1c, r1h, r1w;
    long pre_bstride;
    int H, W, Hv, Wv, ups, Cs0, Cs1, C0;
    int K_pad, nk, Cout, Cout_pad;
    int stride, pad_mode, pre_op, act, store_mode;
    int B, Ho, Wo, M, MT, NT;
    int osy, osx, ooy[4], oox[4];
    int dst_c_off, Cd;
    int ksplit, steps_per_split;
};

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case FUSG_ACT_RELU: return fmaxf(v, 0.f);
        case FUSG_ACT_TANH: return tanhf(v);
        case FUSG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case FUSG_ACT_TANH01: return (tanhf(v) + 1.f) / 2.f;
        default: return v;
    }
}

__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expm1f(v); }

// Pixel part of the destination mapping: offsets (in elements) of logical channel 0 of output
// pixel m in dst / res0 / res1.  Returns false for m >= M.
struct PixOff { long d, r0, r1; };
__device__ __forceinline__ bool pix_offsets(const ConvK& p, int phase, int m, PixOff& o) {
    if (m >= p.M) return false;
    const int hw = p.Ho * p.Wo;
    const int b = m / hw;
    const int rem = m - b * hw;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    long Y, X, cq = 0;
    if (p.store_mode == FUSG_STORE_D2S) { Y = 2 * oy; X = 2 * ox; }
    else if (p.store_mode == FUSG_STORE_S2D) { Y = oy >> 1; X = ox >> 1; cq = (long)(((oy & 1) << 1) | (ox & 1)) * p.Cout; }
    else { Y = (long)oy * p.osy + p.ooy[phase]; X = (long)ox * p.osx + p.oox[phase]; }
    o.d = b * p.dsn + Y * p.dsh + X * p.dsw + (cq + p.dst_c_off) * p.dsc;
    o.r0 = b * p.r0n + Y * p.r0h + X * p.r0w;
    o.r1 = b * p.r1n + Y * p.r1h + X * p.r1w;
    return true;
}
// Channel part.
__device__ __forceinline__ void chan_offsets(const ConvK& p, int n, PixOff& o) {
    if (p.store_mode == FUSG_STORE_D2S) {
        const int q = n / p.Cd;
        const int c = n - q * p.Cd;
        o.d = (long)(q >> 1) * p.dsh + (long)(q & 1) * p.dsw + (long)c * p.dsc;
        o.r0 = 0; o.r1 = 0;
    } else {
        o.d = (long)n * p.dsc; o.r0 = (long)n * p.r0c; o.r1 = (long)n * p.r1c;
    }
}
__device__ __forceinline__ void epi_store(const ConvK& p, const PixOff& po, const PixOff& co, float bias, float v) {
    v = act_apply(v + bias, p.act);
    if (p.res0) v += p.res0[po.r0 + co.r0];
    if (p.res1) v += p.res1[po.r1 + co.r1];
    p.dst[po.d + co.d] = v;
}

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32(const ConvK p) {
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int AP = BM / 32;   // staging passes for the A tile (32 rows x 8 float4 per pass)
    constexpr int BP = BN / 32;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;        // [2][BN][LDK]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kc = t & 7;
    const int r0 = t >> 3;

    // XCD-aware tile order: consecutive tiles (same A rows, different N) stay on one XCD's L2.
    int tile;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, j = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int nt = tile % p.NT;
    const int mt = tile / p.NT;
    const int phase = blockIdx.y;
    const int ks = blockIdx.z;

    // per-thread staging rows
    int rb[AP], riy[AP], rix[AP];
    {
        const int hw = p.Ho * p.Wo;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = mt * BM + r0 + 32 * i;
            if (m < p.M) {
                const int b = m / hw;
                const int rem = m - b * hw;
                const int oy = rem / p.Wo;
                rb[i] = b; riy[i] = oy * p.stride; rix[i] = (rem - oy * p.Wo) * p.stride;
            } else { rb[i] = -1; riy[i] = 0; rix[i] = 0; }
        }
    }
    const int2* ktab = p.ktab + (long)phase * (p.K_pad >> 2);
    const float* wrow = p.wpack + ((long)phase * p.Cout_pad + (long)nt * BN + r0) * p.K_pad + kc * 4;

    f32x4 areg[AP], breg[BP];
    auto load_regs = [&](int s) {
        const int2 e = ktab[s * 8 + kc];
        const int dy = (int)(short)(e.x & 0xffff);
        const int dx = e.x >> 16;
        const bool inval = e.y < 0;
        const int src = (e.y >> 30) & 1;
        const int coff = e.y & 0x3fffffff;
        const float* base = src ? p.src1 : p.src0;
        const int Cs = src ? p.Cs1 : p.Cs0;
        const int cidx = coff + (src ? p.C0 : 0);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        const bool affine = p.pre_op >= FUSG_PRE_AFFINE_RELU;
        if (affine && p.pre_bstride == 0 && !inval) {
            sc = *(const f32x4*)(p.pre_scale + cidx);
            sh = *(const f32x4*)(p.pre_shift + cidx);
        }
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            int iy = riy[i] + dy, ix = rix[i] + dx;
            bool ok = rb[i] >= 0 && !inval;
            if (p.pad_mode == FUSG_PAD_REFLECT) {
                iy = iy < 0 ? -iy : (iy >= p.Hv ? 2 * p.Hv - 2 - iy : iy);
                ix = ix < 0 ? -ix : (ix >= p.Wv ? 2 * p.Wv - 2 - ix : ix);
            } else {
                ok = ok && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
            }
            iy >>= p.ups; ix >>= p.ups;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                v = *(const f32x4*)(base + ((long)(rb[i] * p.H + iy) * p.W + ix) * Cs + coff);
                if (p.pre_op == FUSG_PRE_RELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
                } else if (p.pre_op == FUSG_PRE_ELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = elu1(v[c]);
                } else if (affine) {
                    if (p.pre_bstride != 0) {
                        sc = *(const f32x4*)(p.pre_scale + (long)rb[i] * p.pre_bstride + cidx);
                        sh = *(const f32x4*)(p.pre_shift + (long)rb[i] * p.pre_bstride + cidx);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float y = fmaf(v[c], sc[c], sh[c]);
                        v[c] = p.pre_op == FUSG_PRE_AFFINE_RELU ? fmaxf(y, 0.f) : y;
                    }
                }
            }
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) breg[i] = *(const f32x4*)(wrow + (long)(32 * i) * p.K_pad + s * BK);
    };
    auto store_lds = [&](int buf) {
        float* a = As + buf * BM * LDK + r0 * LDK + kc * 4;
        float* b = Bs + buf * BN * LDK + r0 * LDK + kc * 4;
#pragma unroll
        for (int i = 0; i < AP; ++i) *(f32x4*)(a + 32 * i * LDK) = areg[i];
#pragma unroll
        for (int i = 0; i < BP; ++i) *(f32x4*)(b + 32 * i * LDK) = breg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int s_begin = ks * p.steps_per_split;
    const int s_end = min(p.nk, s_begin + p.steps_per_split);

    if (s_begin < s_end) {
        load_regs(s_begin);
        store_lds(0);
        __syncthreads();
        const int a_off = (wm * TM * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
        const int b_off = (wn * TN * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
        for (int s = s_begin; s < s_end; ++s) {
            const int buf = (s - s_begin) & 1;
            if (s + 1 < s_end) load_regs(s + 1);
            const float* Ab = As + buf * BM * LDK + a_off;
            const float* Bb = Bs + buf * BN * LDK + b_off;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(Ab + i * 32 * LDK + j * 8);
#pragma unroll
                for (int i = 0; i < TN; ++i) b[i] = *(const f32x4*)(Bb + i * 32 * LDK + j * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj)
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[jj][e], acc[i][jj], 0, 0, 0);
            }
            if (s + 1 < s_end) store_lds(buf ^ 1);
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------- epilogue
    const int ncol0 = nt * BN + wn * TN * 32 + (lane & 31);
    const int mrow0 = mt * BM + wm * TM * 32 + 4 * (lane >> 5);
    if (p.ksplit > 1) {
        float* ws = p.ws + ((long)(phase * p.ksplit + ks) * p.M) * p.Cout_pad;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (m < p.M) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) ws[(long)m * p.Cout_pad + ncol0 + j * 32] = acc[i][j][r];
                }
            }
        return;
    }
    PixOff co[TN];
    float bias[TN];
    bool nok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * 32;
        nok[j] = n < p.Cout;
        bias[j] = p.bias[n];
        chan_offsets(p, n, co[j]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
            PixOff po;
            if (pix_offsets(p, phase, m, po)) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (nok[j]) epi_store(p, po, co[j], bias[j], acc[i][j][r]);
            }
        }
}

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
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.ksplit; ++k) {
        const f32x4 v = *(const f32x4*)(p.ws + ((long)(phase * p.ksplit + k) * p.M + m) * p.Cout_pad + c4 * 4);
        s += v;
    }
    PixOff po;
    if (!pix_offsets(p, phase, m, po)) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = c4 * 4 + c;
        if (n < p.Cout) {
            PixOff co;
            chan_offsets(p, n, co);
            epi_store(p, po, co, p.bias[n], s[c]);
        }
    }
}

struct TileCfg { int bm, bn; };
static const TileCfg kTiles[] = {{0, 0}, {128, 128}, {128, 64}, {128, 32}, {64, 64}, {64, 128}};

template <int TM, int TN, int WM, int WN>
static hipError_t launch_cfg(const ConvK& k, dim3 grid, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_f32<TM, TN, WM, WN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_f32<TM, TN, WM, WN>), grid, dim3(256), lds, s, k);
    return hipGetLastError();
}

static int64_t plan_impl(fusg_conv_desc* d) {
    const long M = (long)d->src0.n * d->qh * d->qw;
    const int nphase = d->nphase > 0 ? d->nphase : 1;
    if (d->tile == FUSG_TILE_AUTO) {
        int bn = d->cout_pad >= 128 && d->cout_pad % 128 == 0 ? 128 : (d->cout_pad % 64 == 0 ? 64 : 32);
        int bm = 128;
        // small problems: prefer more, smaller tiles
        long tiles = ((M + 127) / 128) * (d->cout_pad / bn) * nphase;
        if (tiles < 256 && bn == 128) { bn = 64; tiles = ((M + 127) / 128) * (d->cout_pad / bn) * nphase; }
        if (tiles < 256 && bn == 64) { bm = 64; }
        d->tile = bm == 128 ? (bn == 128 ? FUSG_TILE_128x128 : bn == 64 ? FUSG_TILE_128x64 : FUSG_TILE_128x32)
                            : FUSG_TILE_64x64;
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

extern "C" int fusg_conv2d(const fusg_conv_desc* din, void* stream) {
    fusg_conv_desc dd = *din;
    fusg_conv_desc* d = &dd;
    hipStream_t s = (hipStream_t)stream;
    const fusg_tensor& x0 = d->src0;
    FUSG_CHECK(is_nhwc(x0), "conv2d: src0 must be NHWC-physical f32 (sc=1, Cs%%4=0, 16B aligned)");
    const bool has1 = d->src1.data != nullptr;
    if (has1) {
        FUSG_CHECK(is_nhwc(d->src1) && same_nhw(x0, d->src1), "conv2d: src1 must be NHWC-physical with src0's n,h,w");
    }
    FUSG_CHECK(d->wpack && d->bias && d->ktab, "conv2d: wpack/bias/ktab missing");
    FUSG_CHECK((((uintptr_t)d->wpack) & 15) == 0 && (((uintptr_t)d->ktab) & 7) == 0, "conv2d: wpack/ktab misaligned");
    FUSG_CHECK(d->k_pad > 0 && d->k_pad % BK == 0, "conv2d: k_pad %d not a positive multiple of %d", d->k_pad, BK);
    FUSG_CHECK(d->cout > 0 && d->cout_pad % 32 == 0 && d->cout <= d->cout_pad, "conv2d: bad cout %d / cout_pad %d", d->cout, d->cout_pad);
    FUSG_CHECK(d->c0k >= 0 && d->c0k % 4 == 0 && d->c0k <= x0.sw, "conv2d: bad c0k %d (src0 Cs %ld)", d->c0k, (long)x0.sw);
    FUSG_CHECK(d->stride == 1 || d->stride == 2, "conv2d: stride %d", d->stride);
    FUSG_CHECK(d->upsample == 0 || d->upsample == 1, "conv2d: upsample %d", d->upsample);
    FUSG_CHECK(d->pad_mode == FUSG_PAD_ZERO || d->pad_mode == FUSG_PAD_REFLECT, "conv2d: pad_mode");
    FUSG_CHECK(d->pre_op >= 0 && d->pre_op <= FUSG_PRE_AFFINE, "conv2d: pre_op");
    FUSG_CHECK(d->act >= 0 && d->act <= FUSG_ACT_TANH01, "conv2d: act");
    if (d->pre_op >= FUSG_PRE_AFFINE_RELU) {
        FUSG_CHECK(d->pre_scale && d->pre_shift && ((((uintptr_t)d->pre_scale) | ((uintptr_t)d->pre_shift)) & 15) == 0 &&
                   d->pre_bstride % 4 == 0, "conv2d: affine pre-op needs 16B-aligned scale/shift");
    }
    const int nphase = d->nphase > 0 ? d->nphase : 1;
    FUSG_CHECK(nphase == 1 || nphase == 4, "conv2d: nphase %d", nphase);
    FUSG_CHECK(d->qh > 0 && d->qw > 0, "conv2d: empty output grid");
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
            oh = oh > (long)(d->qh - 1) * d->out_sy + d->out_oy[ph] + 1 ? oh : (long)(d->qh - 1) * d->out_sy + d->out_oy[ph] + 1;
            ow = ow > (long)(d->qw - 1) * d->out_sx + d->out_ox[ph] + 1 ? ow : (long)(d->qw - 1) * d->out_sx + d->out_ox[ph] + 1;
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

    ConvK k;
    memset(&k, 0, sizeof(k));
    k.src0 = (const float*)x0.data; k.src1 = has1 ? (const float*)d->src1.data : (const float*)x0.data;
    k.wpack = d->wpack; k.bias = d->bias; k.ktab = (const int2*)d->ktab;
    k.pre_scale = d->pre_scale; k.pre_shift = d->pre_shift; k.pre_bstride = d->pre_bstride;
    k.dst = (float*)o.data; k.dsn = o.sn; k.dsc = o.sc; k.dsh = o.sh; k.dsw = o.sw;
    if (d->res0.data) { k.res0 = (const float*)d->res0.data; k.r0n = d->res0.sn; k.r0c = d->res0.sc; k.r0h = d->res0.sh; k.r0w = d->res0.sw; }
    if (d->res1.data) { k.res1 = (const float*)d->res1.data; k.r1n = d->res1.sn; k.r1c = d->res1.sc; k.r1h = d->res1.sh; k.r1w = d->res1.sw; }
    k.ws = d->workspace;
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

    dim3 grid(k.MT * k.NT, nphase, d->ksplit);
    const double flops = 2.0 * (double)Ml * d->cout * d->k_pad * nphase;   // padded-K flops; bench uses algorithmic ones
    prof_begin(0, s, flops);
    hipError_t e;
    switch (d->tile) {
        case FUSG_TILE_128x128: e = launch_cfg<2, 2, 2, 2>(k, grid, s); break;
        case FUSG_TILE_128x64:  e = launch_cfg<2, 1, 2, 2>(k, grid, s); break;
        case FUSG_TILE_128x32:  e = launch_cfg<1, 1, 4, 1>(k, grid, s); break;
        case FUSG_TILE_64x64:   e = launch_cfg<1, 1, 2, 2>(k, grid, s); break;
        default:                e = launch_cfg<1, 2, 2, 2>(k, grid, s); break;   // 64x128
    }
    if (e != hipSuccess) { set_error("conv2d launch: %s", hipGetErrorString(e)); prof_end(0, s); return FUSG_ERR_LAUNCH; }
    if (d->ksplit > 1) {
        const long total = (long)nphase * k.M * (k.Cout_pad >> 2);
        hipLaunchKernelGGL(conv_splitk_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k, nphase);
        e = hipGetLastError();
        if (e != hipSuccess) { set_error("conv2d split-K reduce launch: %s", hipGetErrorString(e)); prof_end(0, s); return FUSG_ERR_LAUNCH; }
    }
    prof_end(0, s);
    return FUSG_OK;
}
