// One hourglass Bottleneck per launch (fusg_hg_bottleneck, include/fusg.h): split-fp16 arithmetic of conv_kernel_h3.h,
//   out = res + conv3( relu( conv2( relu( conv1( relu(x * s1 + h1) ) + b1 ) ) + b2 ) ) + b3
// conv1: 1x1 Cin -> 128, conv2: 3x3 128 -> 128 (zero padding 1), conv3: 1x1 128 -> 256 (stacked_hourglass/models.py:22-42,
// bn2 / bn3 folded into conv1 / conv2 by the caller).
//
// Why: as three launches the block moves Cin + 5*128 + 3*256 floats per pixel through HBM/L2 and, on the hourglass's
// 4x4 .. 16x16 levels, every launch is a few microseconds of work behind ~8 us of launch-to-launch latency (102 conv
// launches per hourglass pass at 86 TFLOP/s, profiles/r02_layer_profile.txt).  Here a workgroup owns an 8 x 8 patch of
// output pixels of one image and keeps both intermediates in LDS, already split into (hi, lo') fp16 pairs:
//   1. conv1 on the patch's 10 x 10 halo (112 GEMM rows, 100 used; rows outside the image are forced to 0 = conv2's
//      zero padding).  x is staged chunk by chunk (32 channels: affine + ReLU + split -> LDS image S), the next
//      chunk's global loads in flight behind the MFMAs of the current one.
//   2. conv2: 4 chunks x 9 taps straight from the LDS image T of conv1's output.
//   3. conv3 from the LDS image U of conv2's output (overlaid on T), residual added, 16-byte stores.
// The MFMA runs "transposed" - weights as the A operand, pixels as the B operand of v_mfma_f32_16x16x32_f16 (both have
// the same register layout: lane l = row/column l & 15, k 8 (l >> 4) .. +8) - so that a lane ends up with 4 consecutive
// output channels of one pixel: conv1 / conv2 results go to LDS as one 8-byte store per fp16 half, conv3's to memory
// as 16-byte stores next to 16-byte residual loads, with no transposition step.
// Waves split the output channels (no weight fragment is fetched twice per workgroup) and all read every pixel
// fragment from LDS.  LDS images: 64-byte pixels (32 halves), the 16-byte slot g of pixel p stored at slot
// g ^ 2 ((p >> 2) & 1) (S, U: fragments are 16 consecutive pixels) or g ^ 2 (hy & 1) (T: fragments are 2 halo rows x 8
// columns): conflict-free for ds_read_b128's four 16-lane service groups on every tap shift (exhaustive check of the
// group table in MI355X_MICROARCH.md, LDS).  64 KiB per workgroup -> 2 workgroups per CU.
#include "conv_kernel_h3.h"

namespace fusg {

struct BneckK {
    const float* x; const float* res; float* dst;
    long xsn, xsh, xsw, rsn, rsh, rsw, dsn, dsh, dsw;
    const float* pre_scale; const float* pre_shift;
    const _Float16* w1; const _Float16* w2; const _Float16* w3;
    const float* b1; const float* b2; const float* b3;
    const float* s1; const float* s2; const float* s3;
    const float* zeros;
    int* status;
    int B, H, W, Cin, tiles_x, tiles_y;
    int touch_w;                          // 1: the first workgroups warm L2 with the block's three weight panels (l2_touch)
};

constexpr int BN_HP = 100;                // 10 x 10 halo pixels
constexpr int BN_SROWS = 112;             // ... as 7 row groups of 16
constexpr size_t bneck_lds(int planes) { return (size_t)(2 * BN_SROWS * 32 + 2 * (planes / 32) * BN_HP * 32) * sizeof(_Float16); }   // 64 KiB at 128

template <int CT> struct WFrag { h8 f[CT][2]; };      // [16-column tile][hi | lo]

// Diagnostic build only (-DFUSG_BNECK_STAMPS, tools/bneck_stamps.py): s_memtime at the phase boundaries of the first 64 workgroups, into
// a buffer nothing else reads.  The product build executes no stamp.
#ifdef FUSG_BNECK_STAMPS
__device__ unsigned long long g_bneck_stamps[64 * 12];
#define FUSG_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_bneck_stamps[blockIdx.x * 12 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FUSG_STAMP_RT(i) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_bneck_stamps[blockIdx.x * 12 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FUSG_STAMP(i) do {} while (0)
#define FUSG_STAMP_RT(i) do {} while (0)
#endif

__device__ __forceinline__ f32x4 mfma16(const h8 a, const h8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// P = planes (128: every block of the hourglass levels; 64: the `layer1` block).  The four waves split the output
// channels: CT1 = P / 64 16-channel tiles of conv1 / conv2 per wave, CT3 = P / 32 of conv3's 2 P channels.
template <int P>
__global__ __launch_bounds__(256, 2) void hg_bneck_h3(const BneckK k) {
    constexpr int NCH = P / 32;                          // 32-channel chunks of the intermediates
    constexpr int CT1 = P / 64, CT3 = P / 32;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* Sh = smem_h;                              // conv1's operand, one 32-channel chunk: [112][32]
    _Float16* Sl = Sh + BN_SROWS * 32;
    _Float16* Th = Sl + BN_SROWS * 32;                  // conv1's output on the halo: [NCH chunks][100][32]
    _Float16* Tl = Th + NCH * BN_HP * 32;
    _Float16* Uh = Th;                                  // conv2's output on the patch: [NCH chunks][64][32] (overlays T)
    _Float16* Ul = Th + NCH * 64 * 32;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lp = lane & 15, lg = lane >> 4;           // pixel of the fragment, 8-k slot
    FUSG_STAMP_RT(8);
    FUSG_STAMP(0);
    int tile;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, j = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tpi = k.tiles_x * k.tiles_y;
    const int b = tile / tpi, t2 = tile - b * tpi;
    const int ty = t2 / k.tiles_x, tx = t2 - ty * k.tiles_x;
    const int oy0 = ty * 8, ox0 = tx * 8;
    float amax = 0.f;
    const float ninf = -__builtin_inff();

    if (k.touch_w && gridDim.x <= TOUCH_MAX_WGS / 2) {
        // Inside the pass every block meets its ~850 KiB of weights cold, and the loops below fetch them two steps
        // ahead: on the 4 x 4 .. 16 x 16 levels (32 - 128 workgroups) that is a memory round trip per step, 48 steps in
        // a row.  One dword per 128-byte line of all three panels now: conv1's first (needed at once), then conv2's,
        // then conv3's (conv_kernel.h, l2_touch: what it bought, and why larger grids do not do it).
        void* dummy = (char*)smem_h + bneck_lds(P);
        const long n1 = (long)(k.Cin >> 5) * NCH * 4096, n2 = (long)9 * NCH * NCH * 4096, n3 = (long)NCH * 2 * NCH * 4096;   // bytes
        for (long o = (long)t * 128; o < n1; o += 256 * 128) l2_touch((const char*)k.w1 + o, dummy);
        for (long o = (long)t * 128; o < n2; o += 256 * 128) l2_touch((const char*)k.w2 + o, dummy);
        for (long o = (long)t * 128; o < n3; o += 256 * 128) l2_touch((const char*)k.w3 + o, dummy);
    }

    // ------------------------------------------------------------------ conv1 on the halo
    const int kc = t & 7;
    const float* xp[4];
    int soff[4];
    unsigned sval = 0, sexist = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int item = t + 256 * j, pix = item >> 3;
        xp[j] = k.zeros;
        soff[j] = 0;
        if (item < BN_SROWS * 8) {
            sexist |= 1u << j;
            soff[j] = pix * 32 + ((((kc >> 1) ^ (((pix >> 2) & 1) << 1)) << 3) | ((kc & 1) << 2));
            if (pix < BN_HP) {
                const int hy = pix / 10, hx = pix - hy * 10;
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                if ((unsigned)iy < (unsigned)k.H && (unsigned)ix < (unsigned)k.W) {
                    sval |= 1u << j;
                    xp[j] = k.x + (long)b * k.xsn + (long)iy * k.xsh + (long)ix * k.xsw + kc * 4;
                }
            }
        }
    }
    f32x4 hreg[4], sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int c) {
        sc = *(const f32x4*)(k.pre_scale + c * 32 + kc * 4);
        sh = *(const f32x4*)(k.pre_shift + c * 32 + kc * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) hreg[j] = *(const f32x4*)(((sval >> j) & 1u) ? xp[j] + c * 32 : k.zeros);
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = hreg[j];
            const bool ok = (sval >> j) & 1u;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float y = fmaf(v[c], sc[c], sh[c]); v[c] = ok ? y : 0.f; }
            h4 hi, lo;
            split4(v, 0.f, hi, lo, amax);                       // bn1's ReLU = the split's floor
            if ((sexist >> j) & 1u) { *(h4*)(Sh + soff[j]) = hi; *(h4*)(Sl + soff[j]) = lo; }
        }
    };
    // weight fragments of 16-column tiles gct0 .. gct0 + CT - 1 of slab `slab` in a panel of nt32 32-column tiles per slab
    // ([slab][nt32][16-column half][hi | lo][64 lanes][8 halves], pack.frag_f16x3)
    auto load_w = [&](auto& F, const _Float16* w, int slab, int nt32, int gct0) {
        constexpr int CT = sizeof(F.f) / sizeof(F.f[0]);
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int gct = gct0 + j;
            const _Float16* p = w + ((long)((slab * nt32 + (gct >> 1)) * 2 + (gct & 1)) * 2) * 512 + lane * 8;
            F.f[j][0] = *(const h8*)(p);
            F.f[j][1] = *(const h8*)(p + 512);
        }
    };

    f32x4 acc1[7][CT1];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < CT1; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sfrag = lp * 32 + ((lg ^ (((lp >> 2) & 1) << 1)) << 3);      // + rg * 512: this lane's fragment in S / U
    auto compute1 = [&](const WFrag<CT1>& F) {
        h8 bs[CT1];
#pragma unroll
        for (int j = 0; j < CT1; ++j) bs[j] = scale_m11(F.f[j][0]);
#pragma unroll
        for (int i0 = 0; i0 < 7; i0 += 2) {
            const int n = i0 + 1 < 7 ? 2 : 1;
            h8 xh[2], xl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (i < n) { xh[i] = *(const h8*)(Sh + (i0 + i) * 512 + sfrag); xl[i] = *(const h8*)(Sl + (i0 + i) * 512 + sfrag); }
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (i < n) {
#pragma unroll
                        for (int j = 0; j < CT1; ++j)
                            acc1[i0 + i][j] = mfma16(term == 0 ? F.f[j][0] : term == 1 ? F.f[j][1] : bs[j], term == 2 ? xl[i] : xh[i], acc1[i0 + i][j]);
                    }
        }
    };

    const int nch1 = k.Cin >> 5;
    WFrag<CT1> wA, wB;
    FUSG_STAMP(1);
    issue(0);
    load_w(wA, k.w1, 0, NCH, wave * CT1);
    for (int c = 0; c < nch1; c += 2) {
        commit();
        __syncthreads();
        if (c + 1 < nch1) { issue(c + 1); load_w(wB, k.w1, c + 1, NCH, wave * CT1); }
        compute1(wA);
        __syncthreads();                                        // every wave is done with this chunk's image
        if (c + 1 >= nch1) break;
        commit();
        __syncthreads();
        if (c + 2 < nch1) { issue(c + 2); load_w(wA, k.w1, c + 2, NCH, wave * CT1); }
        compute1(wB);
        __syncthreads();
    }

    // first weights of conv2 on their way while conv1's result is written to T
    FUSG_STAMP(2);
    WFrag<CT1> w0, w1, w2;
    load_w(w0, k.w2, 0, NCH, wave * CT1);                       // step 0 = (chunk 0, tap 0)
    load_w(w1, k.w2, NCH, NCH, wave * CT1);                     // step 1 = (chunk 0, tap 1): slab tap * NCH + chunk
    // this lane's 4 channels of 16-channel tile gct: chunk gct >> 1, slot 2 (gct & 1) + (lg >> 1), halves (lg & 1) * 4
    const int sub = (lg & 1) << 2;
    {
#pragma unroll
        for (int j = 0; j < CT1; ++j) {
            const int gct = wave * CT1 + j;
            const int n0 = gct * 16 + lg * 4;
            const f32x4 bias = *(const f32x4*)(k.b1 + n0), wsc = *(const f32x4*)(k.s1 + n0);
            const int g = (gct & 1) * 2 + (lg >> 1);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int pix = i * 16 + lp;
                const int hy = pix / 10, hx = pix - hy * 10;
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                const bool ok = pix < BN_HP && (unsigned)iy < (unsigned)k.H && (unsigned)ix < (unsigned)k.W;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float y = fmaxf(fmaf(acc1[i][j][r], wsc[r], bias[r]), 0.f); v[r] = ok ? y : 0.f; }
                h4 hi, lo;
                split4<false>(v, ninf, hi, lo, amax);
                if (pix < BN_HP) {
                    const int o = ((gct >> 1) * BN_HP + pix) * 32 + ((g ^ ((hy & 1) << 1)) << 3) + sub;
                    *(h4*)(Th + o) = hi;
                    *(h4*)(Tl + o) = lo;
                }
            }
        }
    }
    __syncthreads();
    FUSG_STAMP(3);

    // ------------------------------------------------------------------ conv2: 3x3 over T
    f32x4 acc2[4][CT1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CT1; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int prow = lp >> 3, pcol = lp & 7;                    // the lane's pixel inside a row group: 2 rows x 8 columns
    const int tfrag = (prow * 10 + pcol) * 32;                  // + rg * 640
    auto compute2 = [&](const WFrag<CT1>& F, int c, int tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int o = c * (BN_HP * 32) + tfrag + (ky * 10 + kx) * 32 + ((lg ^ (((prow + ky) & 1) << 1)) << 3);
        h8 xh[4], xl[4], bs[CT1];
#pragma unroll
        for (int i = 0; i < 4; ++i) { xh[i] = *(const h8*)(Th + o + i * 640); xl[i] = *(const h8*)(Tl + o + i * 640); }
#pragma unroll
        for (int j = 0; j < CT1; ++j) bs[j] = scale_m11(F.f[j][0]);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < CT1; ++j)
                    acc2[i][j] = mfma16(term == 0 ? F.f[j][0] : term == 1 ? F.f[j][1] : bs[j], term == 2 ? xl[i] : xh[i], acc2[i][j]);
    };
    // NCH * 9 (chunk, tap) steps, chunk-major; weights two steps ahead in a ring of three fragment sets.
    // The weight slab of step s = c * 9 + tap is tap * NCH + c.
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; tap += 3) {
            const int s2 = c * 9 + tap + 2;                     // steps s2, s2+1, s2+2 are fetched in this round
            auto slab_of = [](int s) { const int cc = s / 9; return (s - cc * 9) * NCH + cc; };
            load_w(w2, k.w2, slab_of(s2), NCH, wave * CT1);
            compute2(w0, c, tap);
            if (s2 + 1 < NCH * 9) load_w(w0, k.w2, slab_of(s2 + 1), NCH, wave * CT1);
            compute2(w1, c, tap + 1);
            if (s2 + 2 < NCH * 9) load_w(w1, k.w2, slab_of(s2 + 2), NCH, wave * CT1);
            compute2(w2, c, tap + 2);
        }
    }
    FUSG_STAMP(4);
    __syncthreads();                                            // T is dead: U may overwrite it

    // conv3's first weights, then conv2's result into U
    WFrag<CT3> v0, v1;
    load_w(v0, k.w3, 0, 2 * NCH, wave * CT3);
    {
#pragma unroll
        for (int j = 0; j < CT1; ++j) {
            const int gct = wave * CT1 + j;
            const int n0 = gct * 16 + lg * 4;
            const f32x4 bias = *(const f32x4*)(k.b2 + n0), wsc = *(const f32x4*)(k.s2 + n0);
            const int g = (gct & 1) * 2 + (lg >> 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(acc2[i][j][r], wsc[r], bias[r]), 0.f);
                h4 hi, lo;
                split4<false>(v, ninf, hi, lo, amax);
                const int o = ((gct >> 1) * 64 + i * 16 + lp) * 32 + ((g ^ (((lp >> 2) & 1) << 1)) << 3) + sub;
                *(h4*)(Uh + o) = hi;
                *(h4*)(Ul + o) = lo;
            }
        }
    }
    __syncthreads();
    FUSG_STAMP(5);

    // ------------------------------------------------------------------ conv3: 1x1 P -> 2 P, this wave's P / 2 channels
    f32x4 acc3[4][CT3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CT3; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute3 = [&](const WFrag<CT3>& F, int c) {
        h8 xh[4], xl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { xh[i] = *(const h8*)(Uh + c * 2048 + i * 512 + sfrag); xl[i] = *(const h8*)(Ul + c * 2048 + i * 512 + sfrag); }
        h8 bs[CT3];
#pragma unroll
        for (int j = 0; j < CT3; ++j) bs[j] = scale_m11(F.f[j][0]);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < CT3; ++j)
                    acc3[i][j] = mfma16(term == 0 ? F.f[j][0] : term == 1 ? F.f[j][1] : bs[j], term == 2 ? xl[i] : xh[i], acc3[i][j]);
    };
#pragma unroll
    for (int c = 0; c < NCH; c += 2) {
        load_w(v1, k.w3, c + 1, 2 * NCH, wave * CT3);
        compute3(v0, c);
        if (c + 2 < NCH) load_w(v0, k.w3, c + 2, 2 * NCH, wave * CT3);
        compute3(v1, c + 1);
    }

    FUSG_STAMP(6);
    if (amax >= F16X3_LIMIT && k.status) *k.status = 1;

    // ------------------------------------------------------------------ + bias, + residual, store
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oy = oy0 + 2 * i + prow, ox = ox0 + pcol;
        if (oy >= k.H || ox >= k.W) continue;
        const float* rp = k.res + (long)b * k.rsn + (long)oy * k.rsh + (long)ox * k.rsw;
        float* dp = k.dst + (long)b * k.dsn + (long)oy * k.dsh + (long)ox * k.dsw;
#pragma unroll
        for (int j = 0; j < CT3; ++j) {
            const int n0 = (wave * CT3 + j) * 16 + lg * 4;
            const f32x4 bias = *(const f32x4*)(k.b3 + n0), wsc = *(const f32x4*)(k.s3 + n0);
            const f32x4 r = *(const f32x4*)(rp + n0);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc3[i][j][e], wsc[e], bias[e]) + r[e];
            *(f32x4*)(dp + n0) = v;
        }
    }
    FUSG_STAMP(7);
    FUSG_STAMP_RT(9);
}


// ---- the same block in EXACT fp32 (fusg_bneck_desc.exact_f32 = 1; round 4): v_mfma_f32_16x16x4_f32, weights as the A
// operand (lane (n = l & 15, g = l >> 4) feeds w[n][16 h + 4 g + e] to instruction (h, e) of a 32-channel chunk: the fp32
// fragment copy of pack.frag_f32, two 16-byte loads per 16-channel tile and chunk), pixels as B (the lane's pixel, channels
// 16 h + 4 g + e: two 16-byte LDS reads per fragment and chunk).  The LDS images hold fp32 - 128 bytes per pixel and chunk,
// the bytes of the (hi, lo) pair - unswizzled: the matrix pipe is 16x slower per FLOP than in fp16 and LDS is idle.
// Same structure, phases, barriers and register tiling as hg_bneck_h3; no weight scales, no range status.
template <int CT> struct WFragF { f32x4 f[CT][2]; };    // [16-channel tile][h]

template <int P>
__global__ __launch_bounds__(256, 2) void hg_bneck_f32(const BneckK k) {
    constexpr int NCH = P / 32;
    constexpr int CT1 = P / 64, CT3 = P / 32;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* S = smem_f;                                   // conv1's operand, one 32-channel chunk: [112][32]
    float* T = S + BN_SROWS * 32;                       // conv1's output on the halo: [NCH][100][32]
    float* U = T;                                        // conv2's output on the patch: [NCH][64][32] (overlays T)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lp = lane & 15, lg = lane >> 4;
    int tile;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, j = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tpi = k.tiles_x * k.tiles_y;
    const int b = tile / tpi, t2 = tile - b * tpi;
    const int ty = t2 / k.tiles_x, tx = t2 - ty * k.tiles_x;
    const int oy0 = ty * 8, ox0 = tx * 8;
    const float* w1 = (const float*)k.w1; const float* w2 = (const float*)k.w2; const float* w3 = (const float*)k.w3;

    // ------------------------------------------------------------------ conv1 on the halo
    const int kc = t & 7;
    const float* xp[4];
    int soff[4];
    unsigned sval = 0, sexist = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int item = t + 256 * j, pix = item >> 3;
        xp[j] = k.zeros;
        soff[j] = 0;
        if (item < BN_SROWS * 8) {
            sexist |= 1u << j;
            soff[j] = pix * 32 + kc * 4;
            if (pix < BN_HP) {
                const int hy = pix / 10, hx = pix - hy * 10;
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                if ((unsigned)iy < (unsigned)k.H && (unsigned)ix < (unsigned)k.W) {
                    sval |= 1u << j;
                    xp[j] = k.x + (long)b * k.xsn + (long)iy * k.xsh + (long)ix * k.xsw + kc * 4;
                }
            }
        }
    }
    f32x4 hreg[4], sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int c) {
        sc = *(const f32x4*)(k.pre_scale + c * 32 + kc * 4);
        sh = *(const f32x4*)(k.pre_shift + c * 32 + kc * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) hreg[j] = *(const f32x4*)(((sval >> j) & 1u) ? xp[j] + c * 32 : k.zeros);
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = hreg[j];
            const bool ok = (sval >> j) & 1u;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float y = fmaxf(fmaf(v[c], sc[c], sh[c]), 0.f); v[c] = ok ? y : 0.f; }   // bn1 + ReLU
            if ((sexist >> j) & 1u) *(f32x4*)(S + soff[j]) = v;
        }
    };
    // fp32 fragments of 16-channel tiles gct0 .. of slab `slab`: [slab][nt32][16-channel half][h][64 lanes][4 floats]
    auto load_w = [&](auto& F, const float* w, int slab, int nt32, int gct0) {
        constexpr int CT = sizeof(F.f) / sizeof(F.f[0]);
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int gct = gct0 + j;
            const float* p = w + ((long)((slab * nt32 + (gct >> 1)) * 2 + (gct & 1)) * 2) * 256 + lane * 4;
            F.f[j][0] = *(const f32x4*)(p);
            F.f[j][1] = *(const f32x4*)(p + 256);
        }
    };
    // one (row group of 16 pixels) x (CT tiles) x (32-channel chunk) contraction: xa / xb = the pixel's channels 4 g .. and 16 + 4 g ..
    auto mac = [&](auto& acc_row, const auto& F, const f32x4 xa, const f32x4 xb) __attribute__((always_inline)) {
        constexpr int CT = sizeof(F.f) / sizeof(F.f[0]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    acc_row[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.f[j][h][e], h ? xb[e] : xa[e], acc_row[j], 0, 0, 0);
    };

    f32x4 acc1[7][CT1];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < CT1; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sfrag = lp * 32 + lg * 4;                          // + rg * 512: this lane's first fragment half in S / U
    auto compute1 = [&](const WFragF<CT1>& F) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const f32x4 xa = *(const f32x4*)(S + i * 512 + sfrag), xb = *(const f32x4*)(S + i * 512 + sfrag + 16);
            mac(acc1[i], F, xa, xb);
        }
    };

    const int nch1 = k.Cin >> 5;
    WFragF<CT1> wA, wB;
    issue(0);
    load_w(wA, w1, 0, NCH, wave * CT1);
    for (int c = 0; c < nch1; c += 2) {
        commit();
        __syncthreads();
        if (c + 1 < nch1) { issue(c + 1); load_w(wB, w1, c + 1, NCH, wave * CT1); }
        compute1(wA);
        __syncthreads();
        if (c + 1 >= nch1) break;
        commit();
        __syncthreads();
        if (c + 2 < nch1) { issue(c + 2); load_w(wA, w1, c + 2, NCH, wave * CT1); }
        compute1(wB);
        __syncthreads();
    }

    WFragF<CT1> w0, w1f, w2f;
    load_w(w0, w2, 0, NCH, wave * CT1);
    load_w(w1f, w2, NCH, NCH, wave * CT1);
    {   // conv1's result (+ bias, ReLU; zero outside the image = conv2's padding) -> T
#pragma unroll
        for (int j = 0; j < CT1; ++j) {
            const int gct = wave * CT1 + j;
            const int n0 = gct * 16 + lg * 4;
            const f32x4 bias = *(const f32x4*)(k.b1 + n0);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int pix = i * 16 + lp;
                const int hy = pix / 10, hx = pix - hy * 10;
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                const bool ok = pix < BN_HP && (unsigned)iy < (unsigned)k.H && (unsigned)ix < (unsigned)k.W;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float y = fmaxf(acc1[i][j][r] + bias[r], 0.f); v[r] = ok ? y : 0.f; }
                if (pix < BN_HP) *(f32x4*)(T + ((gct >> 1) * BN_HP + pix) * 32 + (gct & 1) * 16 + lg * 4) = v;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ conv2: 3x3 over T
    f32x4 acc2[4][CT1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CT1; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int prow = lp >> 3, pcol = lp & 7;
    const int tfrag = (prow * 10 + pcol) * 32 + lg * 4;          // + rg * 640
    auto compute2 = [&](const WFragF<CT1>& F, int c, int tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int o = c * (BN_HP * 32) + tfrag + (ky * 10 + kx) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 xa = *(const f32x4*)(T + o + i * 640), xb = *(const f32x4*)(T + o + i * 640 + 16);
            mac(acc2[i], F, xa, xb);
        }
    };
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; tap += 3) {
            const int s2 = c * 9 + tap + 2;
            auto slab_of = [](int s) { const int cc = s / 9; return (s - cc * 9) * NCH + cc; };
            load_w(w2f, w2, slab_of(s2), NCH, wave * CT1);
            compute2(w0, c, tap);
            if (s2 + 1 < NCH * 9) load_w(w0, w2, slab_of(s2 + 1), NCH, wave * CT1);
            compute2(w1f, c, tap + 1);
            if (s2 + 2 < NCH * 9) load_w(w1f, w2, slab_of(s2 + 2), NCH, wave * CT1);
            compute2(w2f, c, tap + 2);
        }
    }
    __syncthreads();                                            // T is dead: U may overwrite it

    WFragF<CT3> v0, v1;
    load_w(v0, w3, 0, 2 * NCH, wave * CT3);
    {
#pragma unroll
        for (int j = 0; j < CT1; ++j) {
            const int gct = wave * CT1 + j;
            const int n0 = gct * 16 + lg * 4;
            const f32x4 bias = *(const f32x4*)(k.b2 + n0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc2[i][j][r] + bias[r], 0.f);
                // pixel i * 16 + lp of the patch in U's order = fragment order of conv2's row groups (2 rows x 8 columns)
                *(f32x4*)(U + ((gct >> 1) * 64 + i * 16 + lp) * 32 + (gct & 1) * 16 + lg * 4) = v;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ conv3: 1x1 P -> 2 P
    f32x4 acc3[4][CT3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CT3; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute3 = [&](const WFragF<CT3>& F, int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 xa = *(const f32x4*)(U + c * 2048 + i * 512 + sfrag), xb = *(const f32x4*)(U + c * 2048 + i * 512 + sfrag + 16);
            mac(acc3[i], F, xa, xb);
        }
    };
#pragma unroll
    for (int c = 0; c < NCH; c += 2) {
        load_w(v1, w3, c + 1, 2 * NCH, wave * CT3);
        compute3(v0, c);
        if (c + 2 < NCH) load_w(v0, w3, c + 2, 2 * NCH, wave * CT3);
        compute3(v1, c + 1);
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oy = oy0 + 2 * i + prow, ox = ox0 + pcol;
        if (oy >= k.H || ox >= k.W) continue;
        const float* rp = k.res + (long)b * k.rsn + (long)oy * k.rsh + (long)ox * k.rsw;
        float* dp = k.dst + (long)b * k.dsn + (long)oy * k.dsh + (long)ox * k.dsw;
#pragma unroll
        for (int j = 0; j < CT3; ++j) {
            const int n0 = (wave * CT3 + j) * 16 + lg * 4;
            const f32x4 bias = *(const f32x4*)(k.b3 + n0);
            const f32x4 r = *(const f32x4*)(rp + n0);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc3[i][j][e] + bias[e]) + r[e];
            *(f32x4*)(dp + n0) = v;
        }
    }
}

}  // namespace fusg

using namespace fusg;

static int bneck_impl(const fusg_bneck_desc* d, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    FUSG_CHECK(d != nullptr, "hg_bottleneck: null descriptor");
    const fusg_tensor& x = d->x;
    const int P = d->planes;
    FUSG_CHECK(P == 128 || P == 64, "hg_bottleneck: planes %d (64 and 128 are built)", d->planes);
    FUSG_CHECK(is_nhwc(x) && x.c % 32 == 0 && x.c >= 32 && x.c <= 4096, "hg_bottleneck: x must be NHWC-physical f32 with c %% 32 == 0");
    FUSG_CHECK(is_nhwc(d->res) && is_nhwc(d->dst) && same_nhw(x, d->res) && same_nhw(x, d->dst) && d->res.c == 2 * P &&
               d->dst.c == 2 * P, "hg_bottleneck: res / dst must be NHWC-physical f32 [n, 2 * planes, h, w] like x");
    FUSG_CHECK(x.data != d->dst.data && d->res.data != d->dst.data, "hg_bottleneck: dst must not alias x or res (neighbouring patches read x's halo)");
    const bool f32 = d->exact_f32 == 1;
    FUSG_CHECK(d->exact_f32 == 0 || d->exact_f32 == 1, "hg_bottleneck: exact_f32 %d", d->exact_f32);
    const void* ptrs[] = {d->pre_scale, d->pre_shift, d->w1frag, d->bias1, d->w2frag, d->bias2, d->w3frag, d->bias3};
    for (const void* p : ptrs) FUSG_CHECK(p && (((uintptr_t)p) & 15) == 0, "hg_bottleneck: a parameter array is missing or not 16-byte aligned");
    if (!f32) {
        const void* sp[] = {d->wscale1, d->wscale2, d->wscale3};
        for (const void* p : sp) FUSG_CHECK(p && (((uintptr_t)p) & 15) == 0, "hg_bottleneck: a weight-scale array is missing or not 16-byte aligned");
        FUSG_CHECK(d->status != nullptr, "hg_bottleneck: status word missing");
    }
    FUSG_CHECK(x.n * x.h * x.w * x.sw < (1L << 40) && x.n >= 1 && x.h >= 1 && x.w >= 1, "hg_bottleneck: extent");
    BneckK k;
    memset(&k, 0, sizeof(k));
    k.x = (const float*)x.data; k.res = (const float*)d->res.data; k.dst = (float*)d->dst.data;
    k.xsn = x.sn; k.xsh = x.sh; k.xsw = x.sw;
    k.rsn = d->res.sn; k.rsh = d->res.sh; k.rsw = d->res.sw;
    k.dsn = d->dst.sn; k.dsh = d->dst.sh; k.dsw = d->dst.sw;
    k.pre_scale = d->pre_scale; k.pre_shift = d->pre_shift;
    k.w1 = (const _Float16*)d->w1frag; k.w2 = (const _Float16*)d->w2frag; k.w3 = (const _Float16*)d->w3frag;
    k.b1 = d->bias1; k.b2 = d->bias2; k.b3 = d->bias3;
    k.s1 = d->wscale1; k.s2 = d->wscale2; k.s3 = d->wscale3;
    k.status = d->status;
    k.zeros = zero_line();
    if (!k.zeros) { set_error("hg_bottleneck: cannot allocate the zero line"); return FUSG_ERR_LAUNCH; }
    k.B = (int)x.n; k.H = (int)x.h; k.W = (int)x.w; k.Cin = (int)x.c;
    k.tiles_x = (k.W + 7) / 8; k.tiles_y = (k.H + 7) / 8;
    const long wgs = (long)k.B * k.tiles_x * k.tiles_y;
    FUSG_CHECK(wgs < (1L << 31), "hg_bottleneck: grid");
    const void* fn = f32 ? (P == 128 ? (const void*)hg_bneck_f32<128> : (const void*)hg_bneck_f32<64>)
                         : (P == 128 ? (const void*)hg_bneck_h3<128> : (const void*)hg_bneck_h3<64>);
    const size_t lds = bneck_lds(P) + TOUCH_LDS_BYTES;
    k.touch_w = env_switches().no_touch ? 0 : 1;
    if (hipError_t e = ensure_dyn_lds(fn, (int)lds); e != hipSuccess) { set_error("hg_bottleneck: %s", hipGetErrorString(e)); return FUSG_ERR_LAUNCH; }
    const double M = (double)x.n * x.h * x.w;
    prof_begin(0, s, 2.0 * M * ((double)x.c * P + 9.0 * P * P + 2.0 * P * P));   // the three convs' own FLOPs (no halo recompute)
    void* args[] = {(void*)&k};
    const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)wgs), dim3(256), args, lds, s);
    prof_end(0, s);
    if (e != hipSuccess) { set_error("hg_bottleneck launch: %s", hipGetErrorString(e)); return FUSG_ERR_LAUNCH; }
    note_conv_kernel(FUSG_CONV_BNECK);
    return FUSG_OK;
}
extern "C" int fusg_hg_bottleneck(const fusg_bneck_desc* d, void* stream) { return fusg::plan_dispatch(bneck_impl, stream, d); }
#ifdef FUSG_BNECK_STAMPS
extern "C" int fusg_debug_bneck_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fusg::g_bneck_stamps), sizeof(unsigned long long) * 64 * 12) == hipSuccess ? 0 : 1;
}
#endif
