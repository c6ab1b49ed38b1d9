// Split-fp16 convolution for layers with very few input channels (the 7x7 stems: ICN 21, EdgeConnect 3 / 4,
// hourglass 3 channels; stride 1 or 2).
//
// The halo kernel walks K as (tap, 32-channel chunk): with 21 channels a third of its MFMAs multiply padding,
// and with 3-4 channels it does not apply at all, which leaves the generic gather (49 taps x one 16-byte load
// per row and tap: 70-100 TFLOP/s on layers that should run at the HBM rate).  Here the whole halo of the
// 8 x 16 output patch - every channel, padded only to a multiple of 4 - is staged ONCE in LDS as fp16 (hi, lo),
// and K is walked in *units* of U = 8 (channels % 8 == 0) or 4 halves: K index = (tap, unit of the pixel), 16/U
// units per MFMA k-step.  A lane's 8 k-values of a step are one (U = 8) or two (U = 4) units, each a contiguous
// LDS read at pixel offset + uoff[unit]; the table of unit offsets comes in through the kernel arguments.
// ICN stem: 49 taps x 3 units = 147 units = 74 k-steps instead of 98; EdgeConnect / hourglass stems: 13 k-steps.
// Weights: pack.py stores them per k-step in MFMA-fragment order (frag_tapunit), zero where a unit is padding.
// MODE 1 (round 4, FUSG_PREC_BF16: the stems of BASELINE configs[4]'s "bf16 MFMA conv path"): the same kernel with ONE bf16 product per
// operand pair - activations rounded to bf16 while they are staged (one LDS image), weights pre-rounded (pack.py: frag_tapunit_bf16,
// [step][cout_pad/32][64 lanes][8]), v_mfma_f32_32x32x16_bf16, no weight scales, no range status.
#pragma once
#include "conv_kernel_halo.h"

namespace fusg {

struct TapUnitK {
    ConvK c;
    int HH, HW;                 // halo extent in input pixels
    int PP, RP;                 // LDS pitch of a pixel / of a halo row, in halves
    int CP;                     // staged channels per pixel (multiple of 4, <= 24)
    int pad_h, pad_w, stride;
    int tiles_x, tiles_per_img;
    int nunits, nsteps;         // real units, k-steps = ceil(nunits / (16 / U))
    const _Float16* wfrag;      // [step][cout_pad/32][hi|lo][64 lanes][8] halves (MODE 1: [step][cout_pad/32][64 lanes][8] bf16)
    int nt32;
    int uoff[160];              // LDS offset (halves) of each unit relative to the output pixel's halo origin
};

template <int TM, int TN, int WM, int WN, int PK, int U, int MODE = 0>
__global__ __launch_bounds__(256, 2) void conv_tapunit_h3(const TapUnitK hk) {
    constexpr bool BF = MODE == 1;
    const ConvK& p = hk.c;
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    static_assert(BM == 128 && WM * WN == 4 && (U == 8 || U == 4), "8x16 pixel patch, 4 waves");
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* Ah = smem_h;                         // [HH][RP]
    _Float16* Al = Ah + hk.HH * hk.RP;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;

    int tile;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, j = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int nt = tile % p.NT;
    const int mt = tile / p.NT;
    const int b = mt / hk.tiles_per_img;
    const int t2 = mt - b * hk.tiles_per_img;
    const int ty = t2 / hk.tiles_x, tx = t2 - ty * hk.tiles_x;
    const int oy0 = ty * 8, ox0 = tx * 16;

    // ---- stage the halo: every (pixel, 4-channel group) item is loaded, pre-processed, split and stored once.
    // All of a thread's loads (<= 8) are issued before the first is consumed.
    {
        constexpr int NI = 8;
        const int ipp = hk.CP >> 2;                                // items per pixel
        const int nitems = hk.HH * hk.HW * ipp;                    // <= 256 * NI (checked on the host)
        const float vfloor = (PK != PK_ELU && p.pre_relu) ? 0.f : -__builtin_inff();
        float amax = 0.f;
        const long img_pix0 = (long)b * p.H * p.W;
        f32x4 hreg[NI];
        int hoff[NI];
        unsigned hvalid = 0;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int item = t + 256 * j;
            hoff[j] = -1;
            hreg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (item < nitems) {
                const int pix = item / ipp, kc = item - pix * ipp;
                const int hy = pix / hk.HW, hx = pix - hy * hk.HW;
                int vy = oy0 * hk.stride - hk.pad_h + hy, vx = ox0 * hk.stride - hk.pad_w + hx;
                bool ok = true;
                if (p.pad_mode == FUSG_PAD_ZERO) {
                    ok = (unsigned)vy < (unsigned)p.H && (unsigned)vx < (unsigned)p.W;
                } else if (p.pad_mode == FUSG_PAD_REFLECT) {
                    vy = vy < 0 ? -vy : (vy >= p.H ? 2 * p.H - 2 - vy : vy);
                    vx = vx < 0 ? -vx : (vx >= p.W ? 2 * p.W - 2 - vx : vx);
                }
                vy = min(max(vy, 0), p.H - 1); vx = min(max(vx, 0), p.W - 1);       // (also the replicate mode)
                hoff[j] = hy * hk.RP + hx * hk.PP + kc * 4;
                if (ok) hvalid |= 1u << j;
                const float* ptr = ok ? p.src0 + (img_pix0 + (long)vy * p.W + vx) * p.Cs0 + kc * 4 : p.zeros;
                hreg[j] = *(const f32x4*)ptr;
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (hoff[j] < 0) continue;
            f32x4 v = hreg[j];
            if (PK == PK_ELU) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = elu1(v[c]);
            } else if (PK == PK_AFFINE) {
                const int kc4 = ((t + 256 * j) % ipp) * 4;          // channel offset of the item
                const long o = (long)b * p.pre_bstride + kc4;
                const f32x4 sc = *(const f32x4*)(p.pre_scale + o), sh = *(const f32x4*)(p.pre_shift + o);
                const bool ok = (hvalid >> j) & 1u;
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float y = fmaf(v[c], sc[c], sh[c]); v[c] = ok ? y : 0.f; }
            }
            if constexpr (BF) {
                if (vfloor == 0.f) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = __builtin_fmaxf(v[c], 0.f);
                }
                *(bf4*)(Ah + hoff[j]) = __builtin_convertvector(v, bf4);
            } else {
                h4 hi, lo;
                if constexpr (PK == PK_ELU) split4<false>(v, vfloor, hi, lo, amax); else split4(v, vfloor, hi, lo, amax);
                *(h4*)(Ah + hoff[j]) = hi;
                *(h4*)(Al + hoff[j]) = lo;
            }
        }
        if constexpr (!BF) report_range(p, amax);
    }

    // this wave's weight fragments: column tiles (nt*BN/32 + wn*TN + j), j < TN
    constexpr int FPT = BF ? 1 : 2;                                // fragments per 32-column tile and k-step: [hi | lo] or one bf16
    const _Float16* wfr = hk.wfrag + ((long)(nt * (BN / 32) + wn * TN) * FPT * 64 + lane) * 8;
    const long wstep = (long)hk.nt32 * FPT * 64 * 8;               // 16-bit elements per k-step slab

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int abase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * TM * 32 + i * 32 + (lane & 31);
        abase[i] = (row >> 4) * hk.stride * hk.RP + (row & 15) * hk.stride * hk.PP;
    }
    const int g = lane >> 5;

    struct BFrag { h8 f[TN][FPT]; };
    auto b_load = [&](BFrag& F, int step) {
        const _Float16* base = wfr + (long)step * wstep;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int hl = 0; hl < FPT; ++hl) F.f[j][hl] = *(const h8*)(base + (j * FPT + hl) * 512);
    };
    BFrag bfA, bfB;
    b_load(bfA, 0);
    __syncthreads();                                               // halo staged

    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto compute = [&](int step, const BFrag& F) {
        h8 ah[TM], al[TM];
        // unit indices are wave-uniform per half-wave: fetch both halves' offsets with scalar loads, select by lane half
        if (U == 8) {
            const int ja = step * 2, jb = ja + 1;
            const int offa = hk.uoff[min(ja, hk.nunits - 1)], offb = hk.uoff[min(jb, hk.nunits - 1)];
            const bool live = (g ? jb : ja) < hk.nunits;
            const int off = g ? offb : offa;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = live ? *(const h8*)(Ah + abase[i] + off) : zero8;
                if constexpr (!BF) al[i] = live ? *(const h8*)(Al + abase[i] + off) : zero8;
            }
        } else {
            const int j = step * 4, last = hk.nunits - 1;
            const int o0 = hk.uoff[min(j, last)], o1 = hk.uoff[min(j + 1, last)], o2 = hk.uoff[min(j + 2, last)],
                      o3 = hk.uoff[min(j + 3, last)];
            const bool l0 = (j + 2 * g) < hk.nunits, l1 = (j + 2 * g + 1) < hk.nunits;
            const int oa = g ? o2 : o0, ob = g ? o3 : o1;
            const h4 zero4 = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const h4 a0 = l0 ? *(const h4*)(Ah + abase[i] + oa) : zero4, a1 = l1 ? *(const h4*)(Ah + abase[i] + ob) : zero4;
                ah[i] = h8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                if constexpr (!BF) {
                    const h4 c0 = l0 ? *(const h4*)(Al + abase[i] + oa) : zero4, c1 = l1 ? *(const h4*)(Al + abase[i] + ob) : zero4;
                    al[i] = h8{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                }
            }
        }
        if constexpr (BF) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, ah[i]), __builtin_bit_cast(bf8, F.f[j][0]), acc[i][j], 0, 0, 0);
        } else {
            h8 bs[TN];                                             // wh * 2^-11: B operand of the al' term
#pragma unroll
            for (int j = 0; j < TN; ++j) bs[j] = scale_m11(F.f[j][0]);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al[i] : ah[i],
                                                                           term == 0 ? F.f[j][0] : term == 1 ? F.f[j][FPT - 1] : bs[j],
                                                                           acc[i][j], 0, 0, 0);
        }
    };
    int step = 0;
    for (; step + 1 < hk.nsteps; step += 2) {
        b_load(bfB, step + 1);
        compute(step, bfA);
        if (step + 2 < hk.nsteps) b_load(bfA, step + 2);
        compute(step + 1, bfB);
    }
    if (step < hk.nsteps) compute(step, bfA);

    // ---------------------------------------------------------------- epilogue (same as the halo kernel's)
    const int ncol0 = nt * BN + wn * TN * 32 + (lane & 31);
    if (p.vec_epi) {
        __syncthreads();
        float* wlds = (float*)smem_h + wave * (TM * 32 * TN * 32);
        const int rwave = wm * TM * 32;
        epilogue_vec<TM, TN>(p, wlds, acc, lane, nt * BN + wn * TN * 32, [&](int row, PixOff& po) {
            const int rr = rwave + row;
            pix_offsets_yx(p, b, oy0 + (rr >> 4), ox0 + (rr & 15), po);
            return true;
        }, [&](int i) -> float* {
            return p.stats + ((long)b * p.stats_slots + t2 * 4 + (rwave >> 5) + i) * p.Cout * 2;
        });
        return;
    }
    PixOff co[TN];
    float bias[TN], wsc[TN];
    bool nok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * 32;
        nok[j] = n < p.Cout;
        bias[j] = p.bias[n];
        wsc[j] = p.wscale ? p.wscale[n] : 1.f;
        chan_offsets(p, n, co[j]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            PixOff po;
            pix_offsets_yx(p, b, oy0 + (row >> 4), ox0 + (row & 15), po);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (nok[j]) epi_store(p, po, co[j], bias[j], wsc[j], acc[i][j][r]);
        }
}

inline size_t tapunit_lds_bytes(int HH, int RP) { return (size_t)2 * HH * RP * sizeof(_Float16); }

template <int TM, int TN, int WM, int WN>
hipError_t launch_tapunit(const TapUnitK& k, dim3 grid, hipStream_t s, int pk, int unit, int mode) {       // mode: 0 split-fp16, 1 bf16
    size_t lds = tapunit_lds_bytes(k.HH, k.RP);
    if (lds < (size_t)4 * TM * 32 * TN * 32 * sizeof(float)) lds = (size_t)4 * TM * 32 * TN * 32 * sizeof(float);   // epilogue detour
    if (lds > 80 * 1024) return hipErrorInvalidValue;
    const void* fn = nullptr;
#define FUSG_PICK_U(PKV)                                                                      \
    fn = mode == 1 ? (unit == 8 ? (const void*)conv_tapunit_h3<TM, TN, WM, WN, PKV, 8, 1> : (const void*)conv_tapunit_h3<TM, TN, WM, WN, PKV, 4, 1>) \
                   : (unit == 8 ? (const void*)conv_tapunit_h3<TM, TN, WM, WN, PKV, 8, 0> : (const void*)conv_tapunit_h3<TM, TN, WM, WN, PKV, 4, 0>);
    if (pk == PK_NONE) { FUSG_PICK_U(PK_NONE) } else if (pk == PK_ELU) { FUSG_PICK_U(PK_ELU) } else { FUSG_PICK_U(PK_AFFINE) }
#undef FUSG_PICK_U
    if (hipError_t e = ensure_dyn_lds(fn, 80 * 1024); e != hipSuccess) return e;
    TapUnitK kk = k;
    void* args[] = {(void*)&kk};
    return hipLaunchKernel(fn, grid, dim3(256), args, lds, s);
}

hipError_t launch_tapunit_128(const TapUnitK&, dim3, hipStream_t, int, int, int);
hipError_t launch_tapunit_64(const TapUnitK&, dim3, hipStream_t, int, int, int);
hipError_t launch_tapunit_32(const TapUnitK&, dim3, hipStream_t, int, int, int);

}  // namespace fusg
