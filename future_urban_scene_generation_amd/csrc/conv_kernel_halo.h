// Halo-tiled split-fp16 convolution for k x k layers whose sources have a multiple of 32 channels (the layers that
// carry most of the FLOPs: 3x3 / dilated 3x3 / 5x5(+2x upsample) / 7x1 / 1x1, stride 1; stride-2 k3/k4 in
// parity-quadrant form).
//
// The generic kernel (conv_kernel_h3.h) re-gathers and re-converts the A operand for every tap: for a 3x3 layer every
// activation is fetched from L2, pre-processed (ELU / affine+ReLU) and split into fp16 (hi, lo') nine times, and on
// gfx950 that VALU + L2 work - not the matrix pipe - is what bounds it (measured: MFMA pipe 26 % busy).  Here a
// workgroup owns an 8 x 16 patch of output pixels of ONE image and BN output channels.  For each 32-channel chunk of
// the input it stages the patch's halo ((8+(kh-1)d) x (16+(kw-1)d) pixels) ONCE: gather -> pre-op -> fp16 split ->
// LDS.  All kh*kw taps then read their A fragments from that LDS image; only the (static, pre-split) weight tile of
// each (chunk, tap) streams in.  Per output tile this cuts the A-side L2 traffic and the staging VALU work by ~kh*kw.
//
// Contraction: v_mfma_f32_16x16x32_f16 (16 x 16 tiles, one instruction per 32 k = one staged chunk).  Same FLOPs per
// cycle as 32x32x16, but under the board's power limit the chip holds a higher clock on this shape
// (MI355X_MICROARCH.md, DVFS give-back item 7): measured on the layers of the crop pass, sustained, same card, same
// run: 256->256 3x3 372 -> 412 TFLOP/s, 128->128 3x3 @256 322 -> 348, 5x5+upsample 256->128 402 -> 464, dilated 3x3
// 363 -> 395, joules per launch -8..-13 % (profiles/r02_halo_m16_ab.txt).
// A fragments: one patch row (16 pixels) x 32 channels per instruction: lane l = pixel l & 15, channels
// 8 (l >> 4) .. +8.  The halo image has a 64-byte pixel pitch (no padding) and the 16-byte channel slot g of halo
// column hx is stored at slot g ^ 2 ((hx >> 2) & 1): with that swizzle the four 16-lane service groups of a
// ds_read_b128 each touch 16 distinct 16-byte bank groups for EVERY tap shift (exhaustive search over pitches 64-144 B
// and 4-class swizzles; the unswizzled 64- and 80-byte pitches are 2-way conflicted on this operand shape).
//
// Weights never touch LDS: pack.py stores them a second time in MFMA-fragment order (lane l: column l & 15, k
// 8 (l >> 4) .. +8), so the B operand of every (tap, chunk, 16-column tile) is ONE contiguous 1 KiB wave load (16 B
// per lane) that goes straight into registers, one step ahead of its use.  Waves therefore only meet at chunk
// boundaries (two barriers per kh*kw taps) instead of once per tap.
//
// Loop nest: source -> 32-channel chunk -> tap;  K index of a (chunk, tap) weight tile in the packed
// panel = tap * Ctot + channel (same panels as the generic kernel, no re-packing).
#pragma once
#include <type_traits>
#include "conv_kernel_h3.h"
#ifndef FUSG_HALO_WAVES
#define FUSG_HALO_WAVES 2
#endif

namespace fusg {

// MODE 1 (FUSG_PREC_BF16, BASELINE configs[4]'s "bf16 MFMA conv path"): the same kernel with ONE bf16 product per
// (a, w) pair instead of three fp16 products: operands rounded to bf16 (v_cvt_pk_bf16_f32, ties to even) while the
// halo is staged, one LDS image instead of two, weights pre-rounded to bf16 in fragment order, v_mfma_f32_16x16x32_bf16
// with fp32 accumulation.  bf16 has fp32's exponent range: no scaling, no range guard.  A third of the matrix work and
// half the operand traffic; ~2^-9 relative error per operand - measured on the networks' fixtures: SSIM >= 0.9997 on
// every image output, NOT bit-exact on the hourglass's keypoint argmax, which therefore stays on f16x3
// (tests::test_reduced_precision_evidence, profiles/r02_parity.json).
// MODE 2 (FUSG_PREC_F32, the reference's own arithmetic and the range guard's fallback): the same kernel on the fp32 matrix
// instruction v_mfma_f32_16x16x4_f32 (an exact fp32 fmaf chain, 64 FLOP / clk / SIMD = the fp32 vector peak): the halo image
// holds the pre-processed activations as fp32 (128 bytes per pixel and chunk: the bytes of the (hi, lo) pair), a lane's A
// operand of the 8 instructions of a 32-channel chunk is two 16-byte reads (channels 4g .. 4g+3 and 16 + 4g .. of its pixel,
// g = lane >> 4; instruction (h, e) contracts channels {16h + 4g + e}), and the weights come from a third fragment-order
// copy [tap][chunk32][cout_pad/32][16-column half][h][64 lanes][4 floats] (pack.py: frag_f32) - again one contiguous 1 KiB
// wave load per fragment, the same 4 KiB per 32-column tile and step as the split-fp16 mode.  The matrix pipe is 16x slower
// per FLOP than in fp16, so neither LDS nor the weight stream matters here (16 KiB of reads per 4096 MFMA cycles and wave).
// K runs chunk-major here and tap-major in the generic fp32 kernel: the two agree to fp32 rounding, not bit for bit.
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// Diagnostic build only (-DFUSG_HALO_STAMPS on conv_halo_128.hip, tools/halo_stamps.py): s_memtime of every wave of the first 64 workgroups at
// the boundaries of each chunk (first tap issued / last tap issued / next chunk committed), into a buffer nothing else reads.
#ifdef FUSG_HALO_STAMPS
__device__ unsigned long long g_halo_stamps[64 * 4 * 40];
#define FUSG_HSTAMP(slot) do { if (blockIdx.x < 64 && (slot) < 40 && lane == 0) g_halo_stamps[(blockIdx.x * 4 + wave) * 40 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FUSG_HSTAMP(slot) do {} while (0)
#endif

constexpr int HALO_CH = 32;          // channels per staged chunk = k per MFMA
constexpr int HALO_PP = 32;          // halo pixel pitch in halves (64 B, slots swizzled - see above)

struct HaloK {
    ConvK c;
    int kh, kw, dil, pad_h, pad_w;
    int HH, HW;                 // halo extent in (virtual) input pixels
    int RP;                     // LDS pitch of a halo row in halves (halo_row_pitch)
    int tiles_x, tiles_per_img;
    int c1k;                    // K-channels of src1 (0 if absent)
    const _Float16* wfrag;      // f16x3: [tap][chunk][cout_pad/32][16-column half][hi|lo][64 lanes][8] halves
                                // bf16:  [tap][chunk][cout_pad/32][16-column half][64 lanes][8] bf16
    const int* tile_list;       // optional: patch indices (within an image) to compute; MT = B * tile_count
    int tile_count;
    // Stride-2 k3/k4 pad-1 layers as four stride-1 convolutions of the parity sub-images x[2Y+i, 2X+j] (q = 2i+j):
    // a chunk is then (quadrant, 32 channels) with its own short tap list; LDS reads stay unit-stride.
    int s2d;                    // 1: quadrant form of a stride-2 layer (HH x HW = 10 x 18 sub-image pixels)
    int qtaps[4];               // taps of each quadrant
    int qwoff[4];               // first weight slab of each quadrant (slabs of a quadrant are consecutive)
    int qtdy[4][4], qtdx[4][4]; // halo pixel offset (rows, columns) of each (quadrant, local tap)
    int nt32;                   // cout_pad / 32
    int touch_off;              // byte offset of the L2-touch dummy region in dynamic LDS (conv_kernel.h, l2_touch)
    // Reciprocals of the prologue's divisors (launch_halo fills them): n / d == umulhi(n, ceil(2^32 / d)) for every n the
    // kernel divides (n * d < 2^32, checked on the host, which refuses the launch otherwise; 0 = the divisor is 1).  An integer division costs ~25 VALU
    // instructions on gfx950 and the prologue had 3 + NI of them per thread - a sixth of all VALU work of a 32-column
    // launch, which is VALU-issue-bound (profiles/r03_pmc_narrow_layers.txt).
    unsigned m_hw, m_nt, m_tpi, m_tx;
};

__device__ __forceinline__ int fdiv(int n, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n; }     // m == 0: d == 1

__device__ __forceinline__ void pix_offsets_yx(const ConvK& p, int b, int oy, int ox, PixOff& o) {
    long Y, X, cq = 0;
    if (p.store_mode == FUSG_STORE_D2S) { Y = 2 * oy; X = 2 * ox; }
    else if (p.store_mode == FUSG_STORE_S2D) { Y = oy >> 1; X = ox >> 1; cq = (long)(((oy & 1) << 1) | (ox & 1)) * p.Cout; }
    else { Y = (long)oy * p.osy + p.ooy[0]; X = (long)ox * p.osx + p.oox[0]; }
    o.d = b * p.dsn + Y * p.dsh + X * p.dsw + (cq + p.dst_c_off) * p.dsc;
    o.r0 = b * p.r0n + Y * p.r0h + X * p.r0w;
    o.r1 = b * p.r1n + Y * p.r1h + X * p.r1w;
}

// waves per SIMD the register allocator is asked to allow: the 128-column tile needs ~230 VGPRs (2), the narrower
// tiles far fewer - and their launches (few output channels: VUnet 32 / 64-channel layers, hourglass 1x1) are bound by
// memory latency, so more resident workgroups per CU is what they need
#ifdef FUSG_HALO_OCC
constexpr int halo_waves(int tm, int tn) { return tm * tn == 1 ? 4 : (tm * tn == 2 ? 3 : FUSG_HALO_WAVES); }
#else
constexpr int halo_waves(int, int) { return FUSG_HALO_WAVES; }
#endif

// KS > 1 ("K split over the waves", the narrow column tiles of k x k layers): the workgroup's columns are 32 * WN wide and
// its 4 / WN waves per column each compute ALL 128 pixels of the patch for every KS-th tap of every chunk; the partial
// tiles meet in LDS at the end (summed in wave order: deterministic).  Why: with the M split (round 1-2: 32 pixels x 32
// columns per wave) a (chunk, tap) step is 12 MFMAs = 192 cycles - less than the L2 round trip of the next step's weight
// fragments, and ~60 bookkeeping instructions per step - so those launches ran at an MFMA-pipe busy of 0.22-0.30
// (profiles/r03_pmc_narrow_layers.txt); here a step is 48 MFMAs like on the 128-column tile, every wave fetches different
// weight fragments, and the per-step overhead is paid a quarter as often.  It pays on small grids only (conv_igemm.hip).
// bf16 mode: THREE (FUSG_BF16_OCC=4: four) waves per SIMD.  A bf16 step is 16 MFMAs = 256 cycles of matrix pipe per wave and ~1000 cycles of
// wave time (tools/halo_stamps.py: the gaps are the step's own instruction stream and, once per chunk, the wait behind the next chunk's halo
// gather - loads return in order); two waves per SIMD leave the pipe half idle inside the taps, a third fills it.  Needs <= 168 VGPRs (the
// instantiations that would spill keep two waves) and the epilogue's LDS detour in two halves (32 KiB per workgroup instead of 64).
#ifndef FUSG_BF16_OCC
#define FUSG_BF16_OCC 3
#endif
constexpr bool halo_bf16_dense(int tm, int wm, int pk, int ni, int mode, int ks) {
    return FUSG_BF16_OCC > 2 && mode == 1 && ks == 1 && 32 * tm * wm == 128 &&
           !(tm == 4 && ((pk == PK_AFFINE && ni >= 8) || (pk == PK_NONE && ni >= 10)));      // (these spill at 168 registers)
}
constexpr int halo_waves_mode(int tm, int tn, int wm, int pk, int ni, int mode, int ks) {
    return halo_bf16_dense(tm, wm, pk, ni, mode, ks) ? FUSG_BF16_OCC : halo_waves(tm, tn);
}
template <int TM, int TN, int WM, int WN, int PK, int NI, int MODE, int KS = 1>
__global__ __launch_bounds__(256, (32 * TM * WM == 256) ? 1 : halo_waves_mode(TM, TN, WM, PK, NI, MODE, KS)) void conv_halo_h3(const HaloK hk) {
    constexpr bool BF = MODE == 1, F32 = MODE == 2;
    constexpr int CH = HALO_CH, HPITCH = HALO_PP;
    constexpr int CPP = CH / 4;                    // 16-byte fp32 items per halo pixel
    constexpr int LOGC = 3;
    const ConvK& p = hk.c;
    constexpr int BM = 32 * TM * WM;               // 128 output pixels = 8 rows x 16 columns
    constexpr int BN = 32 * TN * WN;
    constexpr int PR = BM / 16;                    // patch rows
    // BM == 256 ("big patch", bf16 mode only, launch_halo_big): a 16 x 16 pixel patch per workgroup - twice the MFMAs per weight
    // fragment fetched and per step of bookkeeping, a halo of 1.27x instead of 1.41x the patch
    static_assert((BM == 128 || (BM == 256 && MODE == 1 && KS == 1)) && WM * WN * KS == 4, "8x16 (or 16x16) pixel patch, 4 waves");
    static_assert(KS == 1 || (WM == 1 && TN == 1 && TM == 4), "K split: every wave owns the whole patch and one 32-column tile");
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    const int HP = hk.HH * hk.HW;
    _Float16* Ah = smem_h;                         // [HH][RP]: rows of HW pixels x HPITCH halves (+ row padding)
    _Float16* Al = Ah + hk.HH * hk.RP;
    float* Af = (float*)smem_h;                    // MODE 2: [HH][HW] pixels x 32 fp32 (the same bytes as the two fp16 images)

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    FUSG_HSTAMP(0);
    const int wk = wave / (WM * WN);               // K-split index (0 when KS == 1)
    const int wm = (wave % (WM * WN)) / WN, wn = wave % WN;
    const int kc = t & (CPP - 1);

    int tile;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, j = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int mt = fdiv(tile, hk.m_nt);
    const int nt = tile - mt * p.NT;
    int b, t2;
    if (hk.tile_list) { b = mt / hk.tile_count; t2 = hk.tile_list[mt - b * hk.tile_count]; }
    else { b = fdiv(mt, hk.m_tpi); t2 = mt - b * hk.tiles_per_img; }
    const int ty = fdiv(t2, hk.m_tx), tx = t2 - ty * hk.tiles_x;
    const int oy0 = ty * PR, ox0 = tx * 16;

    // ---- halo items of this thread: pixel index inside the source image + validity (same for every chunk)
    int hpix[NI];                                   // iy * W + ix of the (reflected / clamped) source pixel
    int hoff[NI];                                   // LDS offset (halves) of the item
    unsigned hvalid = 0;                            // bit j: item j is inside the image (zero padding)
    unsigned hexist = 0;                            // bit j: item j is a real halo item (j-th pass may overrun HP)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int item = t + 256 * j;
        const int pix = item >> LOGC;
        hpix[j] = 0;
        hoff[j] = 0;
        if (pix < HP) {
            hexist |= 1u << j;
            const int hy = fdiv(pix, hk.m_hw), hx = pix - hy * hk.HW;
            hoff[j] = F32 ? hy * hk.RP + hx * 32 + ((kc ^ ((hx >> 1) & 7)) << 2)         // floats; 16-byte slot kc swizzled by the column
                          : hy * hk.RP + hx * HPITCH + ((((kc >> 1) ^ (((hx >> 2) & 1) << 1)) << 3) | ((kc & 1) << 2));
            int vy = oy0 - hk.pad_h + hy, vx = ox0 - hk.pad_w + hx;
            bool ok = true;
            if (hk.s2d) {
                // sub-image coordinates; reflection of the full image at -1 / H is a clamp of the sub-image
                int sy = oy0 - 1 + hy, sx = ox0 - 1 + hx;
                const int Hs = p.H >> 1, Ws = p.W >> 1;
                if (p.pad_mode == FUSG_PAD_ZERO) ok = (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;
                sy = min(max(sy, 0), Hs - 1); sx = min(max(sx, 0), Ws - 1);
                if (ok) { hvalid |= 1u << j; hpix[j] = 2 * sy * p.W + 2 * sx; }
                continue;
            }
            if (p.pad_mode == FUSG_PAD_REFLECT) {
                vy = vy < 0 ? -vy : (vy >= p.Hv ? 2 * p.Hv - 2 - vy : vy);
                vx = vx < 0 ? -vx : (vx >= p.Wv ? 2 * p.Wv - 2 - vx : vx);
            } else if (p.pad_mode == FUSG_PAD_REPLICATE) {
                vy = min(max(vy, 0), p.Hv - 1);
                vx = min(max(vx, 0), p.Wv - 1);
            } else {
                ok = (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv;
            }
            if (ok) { hvalid |= 1u << j; hpix[j] = (vy >> p.ups) * p.W + (vx >> p.ups); }
        }
    }
    const long img_pix0 = (long)b * p.H * p.W;

    // this wave's weight fragments: tiles (nt*BN/32 + wn*TN + j), j < TN
    constexpr int FPT = BF ? 2 : 4;                            // fragments (1 KiB) per 32-column tile: [ct] or [ct][hi|lo]
    const _Float16* wfr = hk.wfrag + ((long)(nt * (BN / 32) + wn * TN) * FPT * 64 + lane) * 8;
    const long wstep = (long)hk.nt32 * FPT * 64 * 8;          // 16-bit elements per (tap, chunk) slab
    const float vfloor = (PK != PK_ELU && p.pre_relu) ? 0.f : -__builtin_inff();
    float amax = 0.f;
    const int nchq = p.C0 / CH;                                 // chunks per quadrant (quadrant form)
    const int nch0 = hk.s2d ? 4 * nchq : p.C0 / CH, nch = nch0 + hk.c1k / CH;
    const int nch32 = (p.C0 + hk.c1k) >> 5;
    const int ntaps = hk.kh * hk.kw;

    if (p.touch_w && gridDim.x <= TOUCH_MAX_WGS) {
        // this wave column's weight slabs: TN * FPT KiB contiguous per (tap, chunk) slab, nslabs slabs wstep apart;
        // the WM waves that share the column take every WM-th group of 64 lines
        constexpr int LPS = TN * FPT * 8;                         // 128-byte lines per slab
        const int nslabs = ntaps * nch32;                         // (quadrant form: the quadrants' taps add up to kh * kw)
        void* dummy = (char*)smem_h + hk.touch_off;
        const _Float16* wcol = hk.wfrag + (long)(nt * (BN / 32) + wn * TN) * FPT * 64 * 8;
        for (int L0 = (wave / WN) * 64; L0 < nslabs * LPS; L0 += WM * KS * 64) {     // the WM * KS waves of a column share the work
            const int L = L0 + lane;
            if (L < nslabs * LPS) {
                const int sl = L / LPS, q = L - sl * LPS;
                l2_touch(wcol + (long)sl * wstep + q * 64, dummy);
            }
        }
    }

    // Staging registers of one chunk's halo.  (Tried in round 3 and dropped: a second set, fetching two chunks ahead on
    // the narrow tiles - conv time of the pass 24.5 -> 25.6 ms: those launches are bound by instruction issue, not by
    // bytes in flight, and the second set only adds instructions.)
    // (every lambda below is always_inline: left to the inliner, the NI >= 10 instantiations kept the staging lambdas as
    // real calls and their register sets went to scratch - 336-368 bytes per lane)
    struct HSet { f32x4 r[NI]; f32x4 sc, sh; };
    HSet hA;
    hA.sc = f32x4{1.f, 1.f, 1.f, 1.f};
    hA.sh = f32x4{0.f, 0.f, 0.f, 0.f};

    // (q, cq) = quadrant and chunk within it of chunk cg in the quadrant form (the callers count them up: no division)
    auto halo_issue = [&](HSet& S, int cg, int q, int cq) __attribute__((always_inline)) {
        const bool s1 = cg >= nch0;
        const float* base = s1 ? p.src1 : p.src0;
        const int Cs = s1 ? p.Cs1 : p.Cs0;
        int coff = (s1 ? cg - nch0 : cg) * CH + kc * 4;
        long qpix = img_pix0;
        if (hk.s2d) { coff = cq * CH + kc * 4; qpix += (q >> 1) * p.W + (q & 1); }
        if (PK == PK_AFFINE) {
            const long o = (long)b * p.pre_bstride + (s1 ? p.C0 : 0) + coff;
            S.sc = *(const f32x4*)(p.pre_scale + o);
            S.sh = *(const f32x4*)(p.pre_shift + o);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const float* ptr = base + (qpix + hpix[j]) * Cs + coff;
            if (PK != PK_AFFINE) ptr = ((hvalid >> j) & 1u) ? ptr : p.zeros;
            S.r[j] = *(const f32x4*)ptr;
        }
    };
    auto halo_commit = [&](const HSet& S) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            f32x4 v = S.r[j];
            if (PK == PK_ELU) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = elu1(v[c]);
            } else if (PK == PK_AFFINE) {
                const bool ok = (hvalid >> j) & 1u;
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float y = fmaf(v[c], S.sc[c], S.sh[c]); v[c] = ok ? y : 0.f; }
            }
            if constexpr (F32) {
                // (only a fused ReLU clamps: fmaxf(NaN, -inf) would turn a NaN into -inf, and a NaN must reach the output)
                if (vfloor == 0.f) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = __builtin_fmaxf(v[c], 0.f);
                }
                if ((hexist >> j) & 1u) *(f32x4*)(Af + hoff[j]) = v;
            } else if constexpr (BF) {
                if (vfloor == 0.f) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = __builtin_fmaxf(v[c], 0.f);
                }
                const bf4 hb = __builtin_convertvector(v, bf4);
                if ((hexist >> j) & 1u) *(bf4*)(Ah + hoff[j]) = hb;
            } else {
                h4 hi, lo;
                if constexpr (PK == PK_ELU) split4<false>(v, vfloor, hi, lo, amax); else split4(v, vfloor, hi, lo, amax);
                if ((hexist >> j) & 1u) {
                    *(h4*)(Ah + hoff[j]) = hi;
                    *(h4*)(Al + hoff[j]) = lo;
                }
            }
        }
    };
    struct BFrag { h8 f[TN][2][BF ? 1 : 2]; };        // [32-column tile][16-column half][hi, lo] (bf16: one fragment)
    BFrag bfA, bfB;
    auto b_load = [&](BFrag& F, const _Float16* base) __attribute__((always_inline)) {            // base: this wave column's part of one (tap, chunk) slab
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int hl = 0; hl < (BF ? 1 : 2); ++hl)
                    F.f[j][ct][hl] = *(const h8*)(base + ((j * 2 + ct) * (BF ? 1 : 2) + hl) * 512);
    };

    f32x4 acc[2 * TM][2 * TN];                        // [patch row of the wave][16-column group]
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // row group i of the wave (16 output pixels) = patch row wm*TM*2 + i; lane l reads pixel l & 15 of it
    int abase[2 * TM];
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) abase[i] = (wm * TM * 2 + i) * hk.RP;
    auto compute = [&](int dyp, int dxp, const BFrag& F) __attribute__((always_inline)) {          // (dyp, dxp): halo pixel offset of the tap
        const int hx = (lane & 15) + dxp;
        const int toff = dyp * hk.RP + hx * HPITCH + (((lane >> 4) ^ (((hx >> 2) & 1) << 1)) << 3);
        if constexpr (F32) {
            const int sw = (hx >> 1) & 7, g = lane >> 4;
            const int tf = dyp * hk.RP + hx * 32;
            f32x4 a0[2 * TM], a1[2 * TM];
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) {
                a0[i] = *(const f32x4*)(Af + abase[i] + tf + ((g ^ sw) << 2));
                a1[i] = *(const f32x4*)(Af + abase[i] + tf + (((g + 4) ^ sw) << 2));
            }
            // instruction-major order: consecutive MFMAs write different accumulators (40-cycle dependent latency)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct) {
                                const f32x4 bv = __builtin_bit_cast(f32x4, F.f[j][ct][h]);
                                acc[i][2 * j + ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(h ? a1[i][e] : a0[i][e], bv[e], acc[i][2 * j + ct], 0, 0, 0);
                            }
        } else if constexpr (BF) {
            bf8 ab[2 * TM];
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) ab[i] = *(const bf8*)(Ah + abase[i] + toff);
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[i][2 * j + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[i], __builtin_bit_cast(bf8, F.f[j][ct][0]),
                                                                                       acc[i][2 * j + ct], 0, 0, 0);
        } else {
            h8 ah[2 * TM], al[2 * TM], bs[TN][2];
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) {
                ah[i] = *(const h8*)(Ah + abase[i] + toff);
                al[i] = *(const h8*)(Al + abase[i] + toff);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) bs[j][ct] = scale_m11(F.f[j][ct][0]);   // wh * 2^-11: B operand of the al' term
            // term-major order: consecutive MFMAs write different accumulators (a dependent chain on one accumulator
            // stalls the issue)
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            acc[i][2 * j + ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                term == 2 ? al[i] : ah[i], term == 0 ? F.f[j][ct][0] : term == 1 ? F.f[j][ct][1] : bs[j][ct],
                                acc[i][2 * j + ct], 0, 0, 0);
        }
    };

    // Step bookkeeping without divisions: a step is (chunk, tap); the launch walks chunk-major.  Three cursors - the
    // step being computed, the step whose weights are being fetched (one ahead) and the chunk whose halo is being
    // fetched - each counted up with its quadrant / chunk-in-quadrant (quadrant form) or (ky, kx) (dense tap grid).
    // The weight slab of (tap, chunk) is tap_global * nch32 + chunk32: consecutive taps of a chunk are `tapstride` apart.
    // (Round 2 derived all of this per step with integer divisions: ~350 mostly scalar instructions per step next to 12
    // MFMAs on the 32-column tile - PMC: those launches spent 36 % of their wave cycles issuing and 29 % stalled on issue
    // at an MFMA-pipe busy of 0.22 - 0.30.)
    const long tapstride = (long)nch32 * wstep;
    const int s2d = hk.s2d;
    // computed step
    int cg = 0, tap = 0, ky = 0, kx = 0, qc = 0, cqc = 0, ntc = s2d ? hk.qtaps[0] : ntaps;
    // fetched step (weights)
    int cgn = 0, tapn = 0, qn = 0, cqn = 0, ntn = ntc;
    const _Float16* wnext = wfr + (long)(s2d ? hk.qwoff[0] : 0) * tapstride;
    // fetched chunk (halo)
    int qh = 0, cqh = 0;
    auto next_halo_chunk = [&]() { if (s2d && ++cqh == nchq) { cqh = 0; ++qh; } };

    // weight cursor -> the next (chunk, tap) step
    auto advance_w = [&]() __attribute__((always_inline)) {
        wnext += tapstride;
        if (++tapn == ntn) {
            tapn = 0;
            ++cgn;
            if (s2d) {
                if (++cqn == nchq) { cqn = 0; ++qn; }
                ntn = hk.qtaps[qn & 3];
                wnext = wfr + (long)hk.qwoff[qn & 3] * tapstride + (long)cqn * wstep;
            } else {
                wnext = wfr + (long)cgn * wstep;
            }
        }
    };

    if constexpr (KS > 1) {
        // ---- K split over the waves: this wave's steps are the taps wk, wk + KS, ... of every chunk (dense tap grid only)
        auto wptr = [&](int cgx, int tapx) __attribute__((always_inline)) { return wfr + (long)tapx * tapstride + (long)cgx * wstep; };
        halo_issue(hA, 0, 0, 0);
        if (wk < ntaps) b_load(bfA, wptr(0, wk));
        halo_commit(hA);
        __syncthreads();
        bool odd = false;
        for (int cgx = 0; cgx < nch; ++cgx) {
            if (cgx + 1 < nch) halo_issue(hA, cgx + 1, 0, 0);         // in flight during this chunk's taps
            int kyx = 0, kxx = wk;
            while (kxx >= hk.kw) { kxx -= hk.kw; ++kyx; }
            for (int tapx = wk; tapx < ntaps; tapx += KS) {
                int ntap = tapx + KS, ncg = cgx;                       // this wave's next step: its weights are fetched now
                if (ntap >= ntaps) { ntap = wk; ++ncg; }
                const _Float16* wn_ptr = wptr(ncg, ntap);
                if (!odd) { if (ncg < nch) b_load(bfB, wn_ptr); compute(kyx * hk.dil, kxx * hk.dil, bfA); }
                else { if (ncg < nch) b_load(bfA, wn_ptr); compute(kyx * hk.dil, kxx * hk.dil, bfB); }
                odd = !odd;
                kxx += KS;
                while (kxx >= hk.kw) { kxx -= hk.kw; ++kyx; }
            }
            if (cgx + 1 < nch) {
                __syncthreads();                                       // every wave is done with the old halo
                halo_commit(hA);
                __syncthreads();
            }
        }
    } else {
    // ---- prologue: halo of chunk 0 and the first weight fragments
    // (Tried in round 3 and left off: weights TWO steps ahead through a ring of three fragment sets for the short steps -
    // bf16 mode, 16 MFMAs = 256 cycles per step, and the narrow split-fp16 tiles, 12 - 24 MFMAs.  Their waves spend 52 % /
    // 35 - 38 % of their cycles in s_waitcnt (profiles/r03_stalls_bf16_halo_256.txt, r03_pmc_narrow_layers.txt), yet the
    // longer lead made both slower: bf16 leg conv 18.15 -> 18.6 ms, f16x3 leg 24.6 -> 25.2 ms.  FUSG_HALO_RING3 builds it.)
#ifdef FUSG_HALO_RING3
    constexpr bool RING3 = BF || TM * TN <= 2;
#else
    constexpr bool RING3 = false;
#endif
    BFrag bfC;
    halo_issue(hA, 0, 0, 0);
    b_load(bfA, wnext);
    if constexpr (RING3) { advance_w(); if (cgn < nch) b_load(bfB, wnext); }
    halo_commit(hA);
    __syncthreads();
    auto one_step = [&](const BFrag& use, BFrag& fill) __attribute__((always_inline)) {
        if (tap == 0) FUSG_HSTAMP(1 + 3 * cg);
        if (cg == 1) FUSG_HSTAMP(30 + tap);                        // (diagnostic: every tap of the second chunk)
        if (tap == 0 && cg + 1 < nch) {                            // in flight during all taps of this chunk
            next_halo_chunk();
            halo_issue(hA, cg + 1, qh, cqh);
        }
        if (cg == 1 && tap < 2 && nch <= 6) FUSG_HSTAMP(20 + 4 * tap);   // (slots 20.. belong to the chunk stamps of deeper layers)
        advance_w();                                               // the step fetched now: one (bf16: two) ahead
        if (cgn < nch) b_load(fill, wnext);
        if (cg == 1 && tap < 2 && nch <= 6) FUSG_HSTAMP(21 + 4 * tap);
        int dyp = ky * hk.dil, dxp = kx * hk.dil;                  // halo pixel offset of the tap
        if (s2d) { dyp = hk.qtdy[qc & 3][tap]; dxp = hk.qtdx[qc & 3][tap]; }
        compute(dyp, dxp, use);
        if (cg == 1 && tap < 2 && nch <= 6) FUSG_HSTAMP(22 + 4 * tap);
        ++tap;
        if (++kx == hk.kw) { kx = 0; ++ky; }
        if (tap == ntc) {
            FUSG_HSTAMP(2 + 3 * cg);
            tap = 0; ky = 0; kx = 0;
            if (s2d) { if (++cqc == nchq) { cqc = 0; ++qc; } ntc = hk.qtaps[qc & 3]; }
            if (++cg < nch) {
                __syncthreads();                                   // every wave is done with the old halo
                halo_commit(hA);
                __syncthreads();
            }
            FUSG_HSTAMP(3 * cg);
        }
    };
    if constexpr (RING3) {
        for (;;) {
            one_step(bfA, bfC);
            if (cg >= nch) break;
            one_step(bfB, bfA);
            if (cg >= nch) break;
            one_step(bfC, bfB);
            if (cg >= nch) break;
        }
    } else {
        for (;;) {
            one_step(bfA, bfB);
            if (cg >= nch) break;
            one_step(bfB, bfA);
            if (cg >= nch) break;
        }
    }

    }

    if constexpr (!BF && !F32) report_range(p, amax);
    if constexpr (KS > 1) {
        // ---- K split: partial tiles -> LDS ([wave][128 rows][32 columns] fp32, over the halo images), summed in wave order by
        // the wave that finishes those rows: wave (wk, wn) takes rows 128 / KS * wk ... of column tile wn
        constexpr int RW = 128 / KS, TME = RW / 32;
        __syncthreads();                                               // every wave is done with the halo images
        float* slabs = (float*)smem_h;
        float* mine = slabs + wave * (128 * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(i * 16 + (lane >> 4) * 4 + r) * 32 + j * 16 + (lane & 15)] = acc[i][j][r];
        __syncthreads();
        const int rows0 = RW * wk;
        float* dstw = slabs + wn * (128 * 32) + rows0 * 32;           // column wn's slab of K-split 0: the sum lands here
        for (int e = lane; e < RW * 8; e += 64) {                      // RW rows x 8 float4
            f32x4 sum = *(const f32x4*)(dstw + e * 4);
#pragma unroll
            for (int k2 = 1; k2 < KS; ++k2) sum += *(const f32x4*)(slabs + (k2 * WN + wn) * (128 * 32) + rows0 * 32 + e * 4);
            *(f32x4*)(dstw + e * 4) = sum;
        }
        auto pixk = [&](int row, PixOff& po) {
            const int rr = rows0 + row;
            pix_offsets_yx(p, b, oy0 + (rr >> 4), ox0 + (rr & 15), po);
            return true;
        };
        auto statk = [&](int i) -> float* {
            return p.stats + ((long)b * p.stats_slots + t2 * 4 + (rows0 >> 5) + i) * p.Cout * 2;
        };
        if (p.vec_epi) {
            ResRegs<TME, 1> none;
            epilogue_rows<TME, 1>(p, dstw, lane, nt * BN + wn * 32, pixk, statk, none, false);
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int e = lane; e < RW * 32; e += 64) {                 // scalar path: one (row, column) per lane and pass
                const int row = e >> 5, n = nt * BN + wn * 32 + (e & 31);
                if (n >= p.Cout) continue;
                PixOff po, co;
                pixk(row, po);
                chan_offsets(p, n, co);
                epi_store(p, po, co, p.bias[n], p.wscale ? p.wscale[n] : 1.f, dstw[e]);
            }
        }
        return;
    }
    // ---------------------------------------------------------------- epilogue
    auto pixfn = [&](int row, PixOff& po) {
        const int rr = wm * TM * 32 + row;
        pix_offsets_yx(p, b, oy0 + (rr >> 4), ox0 + (rr & 15), po);
        return true;
    };
    auto statfn = [&](int i) -> float* {                       // slot = (patch index, 32-row group of the patch)
        return p.stats + ((long)b * p.stats_slots + t2 * (BM / 32) + ((wm * TM * 32) >> 5) + i) * p.Cout * 2;
    };
    if (p.vec_epi) {
        constexpr bool HALVES = BM == 256 || (TM >= 4 && halo_bf16_dense(TM, WM, PK, NI, MODE, KS));
        if constexpr (HALVES) {
            // the wave's tile (TM * 32 rows) leaves in two halves through a wave-private LDS region of half the size (64 KiB per
            // workgroup instead of 128: two workgroups still share a CU); residuals are read in the pass, not prefetched
            // (the prefetch set would be 128 more registers)
            constexpr int TH = TM / 2;
            __syncthreads();
            float* wlds = (float*)smem_h + wave * (TH * 32 * TN * 32);
            ResRegs<TH, TN> none;
            auto do_half = [&](auto halfc) __attribute__((always_inline)) {
                constexpr int half = decltype(halfc)::value;           // compile-time: the accumulators stay in registers
                f32x4 sub[2 * TH][2 * TN];
#pragma unroll
                for (int i = 0; i < 2 * TH; ++i)
#pragma unroll
                    for (int j = 0; j < 2 * TN; ++j) sub[i][j] = acc[half * 2 * TH + i][j];
                auto pixh = [&](int row, PixOff& po) { return pixfn(half * TH * 32 + row, po); };
                auto stath = [&](int i) -> float* { return statfn(half * TH + i); };
                epilogue_vec16<TH, TN>(p, wlds, sub, lane, nt * BN + wn * TN * 32, pixh, stath, none, false);
            };
            do_half(std::integral_constant<int, 0>{});
            do_half(std::integral_constant<int, 1>{});
            return;
        } else {
        ResRegs<TM, TN> rr;
        const bool pre = p.res0 != nullptr;
        if (pre) res_prefetch<TM, TN>(p, lane, nt * BN + wn * TN * 32, pixfn, rr);     // in flight across the barrier and the LDS detour
        __syncthreads();
        float* wlds = (float*)smem_h + wave * (TM * 32 * TN * 32);
        epilogue_vec16<TM, TN>(p, wlds, acc, lane, nt * BN + wn * TN * 32, pixfn, statfn, rr, pre);
        return;
        }
    }
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {                     // C/D map of the 16x16 tile: col = lane & 15, row = 4 (lane >> 4) + reg
        const int n = nt * BN + wn * TN * 32 + j * 16 + (lane & 15);
        if (n >= p.Cout) continue;
        PixOff co;
        chan_offsets(p, n, co);
        const float bias = p.bias[n], wsc = p.wscale ? p.wscale[n] : 1.f;
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * TM * 32 + i * 16 + (lane >> 4) * 4 + r;
                PixOff po;
                pix_offsets_yx(p, b, oy0 + (row >> 4), ox0 + (row & 15), po);
                epi_store(p, po, co, bias, wsc, acc[i][j][r]);
            }
    }
}

// LDS pitch of one halo row in halves (rows need no padding: one ds_read_b128 instruction reads one patch row)
inline int halo_row_pitch(int HW) { return HW * HALO_PP; }
inline size_t halo_lds_bytes(int HH, int HW) { return (size_t)2 * HH * halo_row_pitch(HW) * sizeof(_Float16); }
// a thread stages at most 10 16-byte items of a 32-channel chunk: halos of up to 320 pixels
inline bool halo_fits(int HH, int HW) { return HH * HW * 8 <= 2560 && halo_lds_bytes(HH, HW) <= 96 * 1024; }

template <int TM, int TN, int WM, int WN, int KS = 1>
hipError_t launch_halo(const HaloK& k, dim3 grid, hipStream_t s, int pk, int mode) {      // mode: 0 split-fp16, 1 bf16, 2 exact fp32
    const int HP = k.HH * k.HW;
    size_t lds = halo_lds_bytes(k.HH, k.HW);
    size_t epi = (size_t)4 * TM * 32 * TN * 32 * sizeof(float);                                                     // epilogue detour (K split: the four partial tiles)
    {   // bf16 at three waves per SIMD: the tile leaves in two halves (exactly the instantiations halo_bf16_dense() names)
        const int nit = (HP * 8 + 255) / 256 <= 6 ? 6 : ((HP * 8 + 255) / 256 <= 8 ? 8 : 10);
        if (TM >= 4 && halo_bf16_dense(TM, WM, pk, nit, mode, KS)) epi /= 2;
    }
    if (lds < epi) lds = epi;
    const int touch_off = (int)lds;
    lds += TOUCH_LDS_BYTES;
    const int ni = (HP * 8 + 255) / 256;
    if (!halo_fits(k.HH, k.HW)) return hipErrorInvalidValue;
    const void* fn = nullptr;
#define FUSG_PICK_NI(PKV, MD)                                                                     \
    if (ni <= 6) fn = (const void*)conv_halo_h3<TM, TN, WM, WN, PKV, 6, MD, KS>;                  \
    else if (ni <= 8) fn = (const void*)conv_halo_h3<TM, TN, WM, WN, PKV, 8, MD, KS>;             \
    else fn = (const void*)conv_halo_h3<TM, TN, WM, WN, PKV, 10, MD, KS>;
    if (mode == 2) {
        if (pk == PK_NONE) { FUSG_PICK_NI(PK_NONE, 2) } else if (pk == PK_ELU) { FUSG_PICK_NI(PK_ELU, 2) } else { FUSG_PICK_NI(PK_AFFINE, 2) }
    } else if (mode == 1) {
        if (pk == PK_NONE) { FUSG_PICK_NI(PK_NONE, 1) } else if (pk == PK_ELU) { FUSG_PICK_NI(PK_ELU, 1) } else { FUSG_PICK_NI(PK_AFFINE, 1) }
    } else {
        if (pk == PK_NONE) { FUSG_PICK_NI(PK_NONE, 0) } else if (pk == PK_ELU) { FUSG_PICK_NI(PK_ELU, 0) } else { FUSG_PICK_NI(PK_AFFINE, 0) }
    }
#undef FUSG_PICK_NI
    if (hipError_t e = ensure_dyn_lds(fn, 96 * 1024 + TOUCH_LDS_BYTES); e != hipSuccess) return e;
    HaloK kk = k;
    kk.RP = halo_row_pitch(k.HW);
    kk.touch_off = touch_off;
    bool fits = true;
    auto magic = [&fits](long nmax, int d) -> unsigned {       // exact for n <= nmax when nmax * d < 2^32; 0 for d == 1
        if (d < 1 || nmax * d >= (1L << 32)) fits = false;
        return d < 2 ? 0u : (unsigned)(((1UL << 32) + (unsigned long)d - 1) / (unsigned long)d);
    };
    const long ntiles = (long)grid.x;
    kk.m_hw = magic(256L * 10 + 255, k.HW);                    // pix <= (255 + 256 * (NI - 1)) >> 3, bounded loosely
    kk.m_nt = magic(ntiles, k.c.NT);
    kk.m_tpi = magic(ntiles, k.tiles_per_img);
    kk.m_tx = magic((long)k.tiles_per_img, k.tiles_x);
    if (!fits) return hipErrorInvalidValue;                    // > 2^32 / d tiles: not a shape this kernel is dispatched for
    void* args[] = {(void*)&kk};
    return hipLaunchKernel(fn, grid, dim3(256), args, lds, s);
}

// bf16 mode, 16 x 16 pixel patch x 128 columns (BM == 256): the host sets HH / tiles_per_img for 16 patch rows
template <int TM, int TN, int WM, int WN>
hipError_t launch_halo_big(const HaloK& k, dim3 grid, hipStream_t s, int pk) {
    static_assert(32 * TM * WM == 256, "16 x 16 pixel patch");
    const int HP = k.HH * k.HW;
    size_t lds = (size_t)k.HH * halo_row_pitch(k.HW) * sizeof(_Float16);                        // ONE (bf16) halo image
    constexpr size_t EPI = (size_t)4 * (TM / 2) * 32 * TN * 32 * sizeof(float);
    if (lds < EPI) lds = EPI;
    const int touch_off = (int)lds;
    lds += TOUCH_LDS_BYTES;
    if (HP * 8 > 256 * 12 || lds > 80 * 1024) return hipErrorInvalidValue;
    const void* fn = pk == PK_NONE ? (const void*)conv_halo_h3<TM, TN, WM, WN, PK_NONE, 12, 1, 1>
                   : pk == PK_ELU  ? (const void*)conv_halo_h3<TM, TN, WM, WN, PK_ELU, 12, 1, 1>
                                   : (const void*)conv_halo_h3<TM, TN, WM, WN, PK_AFFINE, 12, 1, 1>;
    if (hipError_t e = ensure_dyn_lds(fn, 80 * 1024 + TOUCH_LDS_BYTES); e != hipSuccess) return e;
    HaloK kk = k;
    kk.RP = halo_row_pitch(k.HW);
    kk.touch_off = touch_off;
    bool fits = true;
    auto magic = [&fits](long nmax, int d) -> unsigned {
        if (d < 1 || nmax * d >= (1L << 32)) fits = false;
        return d < 2 ? 0u : (unsigned)(((1UL << 32) + (unsigned long)d - 1) / (unsigned long)d);
    };
    const long ntiles = (long)grid.x;
    kk.m_hw = magic(256L * 12 + 255, k.HW);
    kk.m_nt = magic(ntiles, k.c.NT);
    kk.m_tpi = magic(ntiles, k.tiles_per_img);
    kk.m_tx = magic((long)k.tiles_per_img, k.tiles_x);
    if (!fits) return hipErrorInvalidValue;
    void* args[] = {(void*)&kk};
    return hipLaunchKernel(fn, grid, dim3(256), args, lds, s);
}

hipError_t launch_halo_128(const HaloK&, dim3, hipStream_t, int, int);
hipError_t launch_halo_big22(const HaloK&, dim3, hipStream_t, int);        // 16 x 16 patch, waves 2 x 2 (128 pixels x 64 columns each)
hipError_t launch_halo_big14(const HaloK&, dim3, hipStream_t, int);        // 16 x 16 patch, waves 1 x 4 (256 pixels x 32 columns each)
hipError_t launch_halo_64(const HaloK&, dim3, hipStream_t, int, int);
hipError_t launch_halo_32(const HaloK&, dim3, hipStream_t, int, int);
hipError_t launch_halo_32k(const HaloK&, dim3, hipStream_t, int, int);      // K split over the four waves
hipError_t launch_halo_64k(const HaloK&, dim3, hipStream_t, int, int);      // K split over two waves per 32-column tile

}  // namespace fusg
