// Exact-fp32 form of the tap-unit convolution (conv_kernel_tapunit.h) for the few-channel 7x7 stems under FUSG_PREC_F32
// (round 4): the whole halo of the 8 x 16 output patch - every channel, padded to a multiple of 4 - is staged ONCE in LDS as
// fp32, and K is walked in UNITS of 4 channels of one tap: a unit is one 16-byte LDS read per lane (the pixel's 4 channels at the
// unit's offset) and two v_mfma_f32_32x32x2_f32 (lane half g feeds channel g of the unit to the first and 2 + g to the
// second).  Weights: pack.frag_tapunit_f32, [unit][cout_pad/32][64 lanes][2 floats] with lane = g * 32 + column holding
// w[column][4 unit + g] and w[column][4 unit + 2 + g] - one 8-byte load per lane, unit and 32-column tile.
// Before this the stems were the generic fp32 gather (ICN stem 21 -> 64, 7x7 at 256 x 256, B = 32: 3.8 ms at 73 TFLOP/s).
#pragma once
#include "conv_kernel_tapunit.h"

namespace fusg {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct TapUnitF {
    ConvK c;
    int HH, HW;                 // halo extent in input pixels
    int PP, RP;                 // LDS pitch of a pixel / of a halo row, in floats
    int CP;                     // staged channels per pixel (multiple of 4, <= 24)
    int pad_h, pad_w, stride;
    int tiles_x, tiles_per_img;
    int nunits;                 // taps * CP / 4
    const float* wfrag;         // [unit][cout_pad/32][64 lanes][2]
    int nt32;
    int uoff[320];              // LDS offset (floats) of each unit relative to the output pixel's halo origin (read with scalar loads:
                                // the index is wave-uniform)
};

template <int TM, int TN, int WM, int WN, int PK>
__global__ __launch_bounds__(256, 2) void conv_tapunit_f32(const TapUnitF hk) {
    const ConvK& p = hk.c;
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    static_assert(BM == 128 && WM * WN == 4, "8x16 pixel patch, 4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem_tf[];
    float* Af = smem_tf;                           // [HH][RP]

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

    {   // ---- stage the halo (same items as the split-fp16 form, stored as they are)
        constexpr int NI = 8;
        const int ipp = hk.CP >> 2;
        const int nitems = hk.HH * hk.HW * ipp;                    // <= 256 * NI (checked on the host)
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
                vy = min(max(vy, 0), p.H - 1); vx = min(max(vx, 0), p.W - 1);
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
                const int kc4 = ((t + 256 * j) % ipp) * 4;
                const long o = (long)b * p.pre_bstride + kc4;
                const f32x4 sc = *(const f32x4*)(p.pre_scale + o), sh = *(const f32x4*)(p.pre_shift + o);
                const bool ok = (hvalid >> j) & 1u;
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float y = fmaf(v[c], sc[c], sh[c]); v[c] = ok ? y : 0.f; }
            }
            if (PK != PK_ELU && p.pre_relu) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
            }
            *(f32x4*)(Af + hoff[j]) = v;
        }
    }

    const float* wfr = hk.wfrag + ((long)(nt * (BN / 32) + wn * TN) * 64 + lane) * 2;
    const long wstep = (long)hk.nt32 * 64 * 2;                      // floats per unit slab

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

    constexpr int UB = 4;                                           // units whose weights are fetched together
    struct BFrag { f32x2 f[UB][TN]; };
    auto b_load = [&](BFrag& F, int u0) {
#pragma unroll
        for (int q = 0; q < UB; ++q) {
            const int u = min(u0 + q, hk.nunits - 1);
            const float* base = wfr + (long)u * wstep;
#pragma unroll
            for (int j = 0; j < TN; ++j) F.f[q][j] = *(const f32x2*)(base + j * 128);
        }
    };
    BFrag bfA, bfB;
    b_load(bfA, 0);
    __syncthreads();                                               // halo staged

    auto compute = [&](int u0, const BFrag& F) {
#pragma unroll
        for (int q = 0; q < UB; ++q) {
            const int u = u0 + q;
            if (u < hk.nunits) {                                   // (wave-uniform)
                const int off = hk.uoff[u];
                float a0[TM], a1[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const f32x4 v = *(const f32x4*)(Af + abase[i] + off);
                    a0[i] = g ? v[1] : v[0];
                    a1[i] = g ? v[3] : v[2];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], F.f[q][j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], F.f[q][j][1], acc[i][j], 0, 0, 0);
                    }
            }
        }
    };
    int u = 0;
    for (; u + UB < hk.nunits; u += 2 * UB) {
        b_load(bfB, u + UB);
        compute(u, bfA);
        if (u + 2 * UB < hk.nunits) b_load(bfA, u + 2 * UB);
        compute(u + UB, bfB);
    }
    if (u < hk.nunits) compute(u, bfA);

    // ---------------------------------------------------------------- epilogue (the halo kernel's)
    const int ncol0 = nt * BN + wn * TN * 32 + (lane & 31);
    if (p.vec_epi) {
        __syncthreads();
        float* wlds = smem_tf + wave * (TM * 32 * TN * 32);
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
            const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            PixOff po;
            pix_offsets_yx(p, b, oy0 + (row >> 4), ox0 + (row & 15), po);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (nok[j]) epi_store(p, po, co[j], bias[j], 1.f, acc[i][j][r]);
        }
}

template <int TM, int TN, int WM, int WN>
hipError_t launch_tapunit_f32(const TapUnitF& k, dim3 grid, hipStream_t s, int pk) {
    size_t lds = (size_t)k.HH * k.RP * sizeof(float);
    if (lds < (size_t)4 * TM * 32 * TN * 32 * sizeof(float)) lds = (size_t)4 * TM * 32 * TN * 32 * sizeof(float);   // epilogue detour
    if (lds > 80 * 1024) return hipErrorInvalidValue;
    const void* fn = pk == PK_NONE ? (const void*)conv_tapunit_f32<TM, TN, WM, WN, PK_NONE>
                   : pk == PK_ELU  ? (const void*)conv_tapunit_f32<TM, TN, WM, WN, PK_ELU>
                                   : (const void*)conv_tapunit_f32<TM, TN, WM, WN, PK_AFFINE>;
    if (hipError_t e = ensure_dyn_lds(fn, 80 * 1024); e != hipSuccess) return e;
    TapUnitF kk = k;
    void* args[] = {(void*)&kk};
    return hipLaunchKernel(fn, grid, dim3(256), args, lds, s);
}

hipError_t launch_tapunit_f32_128(const TapUnitF&, dim3, hipStream_t, int);
hipError_t launch_tapunit_f32_64(const TapUnitF&, dim3, hipStream_t, int);
hipError_t launch_tapunit_f32_32(const TapUnitF&, dim3, hipStream_t, int);

}  // namespace fusg
