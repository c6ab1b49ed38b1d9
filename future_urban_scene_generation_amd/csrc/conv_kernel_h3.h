// Split-precision variant of the fused implicit-GEMM convolution (gfx950): fp32 operands are split
// on the fly into fp16 pairs and multiplied on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, fp32
// accumulate) as
//     a*w ~= ah*wh + ah*wl + al'*(wh * 2^-11)        (the al*wl term, <= 2^-22 |a w|, is dropped)
//   ah  = a rounded toward zero to fp16 (11 significant bits)
//   al' = (a - ah) * 2^11 rounded to fp16: the residual is SCALED into the range of ah, so it keeps 11
//         significant bits for every |a| in fp16's normal range [2^-14, 2^15) instead of sinking into
//         fp16 subnormals (round 1 stored it unscaled: an absolute 2^-25 floor, i.e. fp32-class only for
//         |a| ~ 1)
//   wh, wl = the weights times a per-output-channel power of two s_n that puts the largest |w| of the
//         channel in [2^13, 2^14) (pack.split_f16x3), wl = w s_n - wh unscaled (its floor, 2^-24, is
//         2^-37 of the channel's largest weight); the epilogue multiplies by 1/s_n (exact).
// The third term's B operand wh * 2^-11 is formed in registers (v_pk_mul_f16, exact unless a weight is
// 2^-17 of its channel's largest, where the term is negligible).  fp16 x fp16 products are exact in
// fp32 and all three terms go into the same fp32 accumulator.  CPU emulation over operand scales
// 1e-4 .. 1e4 (tools/emu_split.py): 0.3-0.7x the error of an fp32 fmaf chain at K = 2304.
// Range: |a| >= 2^15 cannot be represented (al' would overflow).  Nothing is clamped: every staged
// operand feeds a running max |a|, and a workgroup that saw |a| >= 2^15 (or an infinity) raises
// ConvK::status, on which the caller re-runs the pass in exact fp32 (ops.py).  NaNs propagate through the
// contraction, with ONE documented exception shared by every precision (f32 kernel included): a fused ReLU is
// fmaxf(v, 0) / v_max3, which returns 0 for a NaN operand (IEEE maxNum), where torch's relu keeps the NaN - a NaN
// entering a ReLU-fused layer is zeroed and does not raise the status word (v_max3 ignores it).  A NaN anywhere
// else (no pre-op, ELU, affine without ReLU, weights, residuals) reaches the output as a NaN.
// Cost: 3 fp16 MFMAs per 16 k instead of 8 fp32 MFMAs: 5.3x less matrix-pipe time; 3 VALU ops per
// staged element for the split.  Same gather / pre-op / epilogue / split-K machinery as conv_kernel.h.
// LDS: four fp16 tiles per buffer, [row][32 halves] = 64-byte rows whose four 16-byte chunks are
// XOR-swizzled with (row >> 2) & 3: conflict-free ds_read_b128 operand fetches without padding, so a
// 128x128 workgroup needs exactly 64 KiB and two workgroups share a CU's 160 KiB.
#pragma once
#include "conv_kernel.h"

namespace fusg {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int LDH = 32;                 // LDS row pitch in halves (64 B, XOR-swizzled 16-byte chunks)


typedef __fp16 h2 __attribute__((ext_vector_type(2)));     // type returned by cvt_pkrtz

constexpr float F16X3_LIMIT = 32768.f;   // |a| >= 2^15: outside the split's range -> ConvK::status

// Split 4 fp32 values into fp16 (hi, lo') = (RTZ(a), RTN((a - hi) * 2^11)) after v = max(v, floor)
// (`floor` = 0 for a fused ReLU, -inf otherwise) and fold |v| into the running maximum `amax`.
// hi: v_cvt_pkrtz_f16_f32 (0.5 op / element); lo': v_mul (a * 2^11) + v_fma_mix{lo,hi}_f16, which reads
// hi straight from its fp16 half, computes hi * -2^11 + a * 2^11 exactly in fp32 and rounds once to fp16
// (hipcc has no pattern for it: inline asm); amax: v_max3_f32 with |.| source modifiers.  3 VALU ops per
// element + 1 for the floor.
// FLOOR = false: no clamp at all (the ELU pre-op and the fused Bottleneck's intermediates have none: the four v_max against -inf
// were 9 % of the staging VALU work of an ELU layer, which is what bounds the 32 / 64-column launches - DESIGN.md §9).
template <bool FLOOR = true>
__device__ __forceinline__ void split4(f32x4 v, float floor, h4& hi, h4& lo, float& amax) {
    if constexpr (FLOOR) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = __builtin_fmaxf(v[c], floor);
    }
    const h2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
    const unsigned p01 = __builtin_bit_cast(unsigned, h01), p23 = __builtin_bit_cast(unsigned, h23);
    const float m2048 = -2048.f;
    const float s0 = v[0] * 2048.f, s1 = v[1] * 2048.f, s2 = v[2] * 2048.f, s3 = v[3] * 2048.f;
    unsigned l01, l23;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(p01), "s"(m2048), "v"(s0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(p01), "s"(m2048), "v"(s1));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(p23), "s"(m2048), "v"(s2));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(p23), "s"(m2048), "v"(s3));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v[0]), "v"(v[1]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v[2]), "v"(v[3]));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 hp = {p01, p23}, lp = {l01, l23};
    hi = __builtin_bit_cast(h4, hp);
    lo = __builtin_bit_cast(h4, lp);
}

// wh * 2^-11: the B operand of the al' term (v_pk_mul_f16 x 4)
__device__ __forceinline__ h8 scale_m11(const h8 x) { return x * (_Float16)0x1p-11f; }

// raise ConvK::status when this lane staged an operand outside the split's range (plain store: every writer
// stores the same value)
__device__ __forceinline__ void report_range(const ConvK& p, float amax) {
    if (amax >= F16X3_LIMIT && p.status) *p.status = 1;
}

// Out-of-image lanes of the pre-op kinds with f(0) = 0 read ConvK::zeros instead of being masked after the load.

// 1 MFMA : FUSG_VALU_PER_MFMA VALU instruction groups for the LLVM scheduler (cdna_hip_programming.md T19)
#ifndef FUSG_VALU_PER_MFMA
#define FUSG_VALU_PER_MFMA 8
#endif
#if defined(FUSG_FORCE_INTERLEAVE)
#define FUSG_INTERLEAVE()                                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < 3 * 2 * TM * TN; ++q_) {             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       \
        __builtin_amdgcn_sched_group_barrier(0x002, FUSG_VALU_PER_MFMA, 0);      \
    }
#else
#define FUSG_INTERLEAVE()      /* measured on MI355X: the forced 1:8 interleave is ~5% slower than hipcc's own order */
#endif

template <int TM, int TN, int WM, int WN, int PK, bool GEN>
__global__ __launch_bounds__(256, 2) void conv_igemm_h3(const ConvK p) {
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int AP = BM / 32;
    constexpr int BCH = BN * 4;                    // 16-byte chunks in one fp16 weight tile (BN x 32 halves)
    constexpr int BPL = (BCH + 255) / 256;         // chunks per thread
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* Ah = smem_h;                         // [2][BM][LDH]
    _Float16* Al = Ah + 2 * BM * LDH;
    _Float16* Bh = Al + 2 * BM * LDH;              // [2][BN][LDH]
    _Float16* Bl = Bh + 2 * BN * LDH;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kc = t & 7;
    const int r0 = t >> 3;

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
    const int hw = p.Ho * p.Wo;

    int rb[AP], riy[AP], rix[AP];
    long rowoff[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = mt * BM + r0 + 32 * i;
        if (m < p.M) {
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / p.Wo;
            rb[i] = b; riy[i] = (oy + p.qy0) * p.stride; rix[i] = (rem - oy * p.Wo + p.qx0) * p.stride;
        } else { rb[i] = -1; riy[i] = 0; rix[i] = 0; }
        rowoff[i] = rb[i] < 0 ? 0 : ((long)(rb[i] * p.H + riy[i]) * p.W + rix[i]);
    }
    bool uni_b = true;
    long aff_off = 0;
    if (PK == PK_AFFINE && p.pre_bstride != 0) {
        const int m_lo = mt * BM, m_hi = min(mt * BM + BM - 1, p.M - 1);
        uni_b = (m_lo / hw) == (m_hi / hw);
        aff_off = (long)(m_lo / hw) * p.pre_bstride;
    }
    const int2* ktab = p.ktab + (long)phase * (p.K_pad >> 2);
    // weight panels: hi then lo, each [cout_pad][k_pad] halves
    const _Float16* wh = p.wpack_h + ((long)phase * 2 * p.Cout_pad + (long)nt * BN) * p.K_pad;
    const _Float16* wl = wh + (long)p.Cout_pad * p.K_pad;
    int brow[BPL], bpiece[BPL];
#pragma unroll
    for (int j = 0; j < BPL; ++j) {
        const int q = t + 256 * j;
        brow[j] = q >> 2;
        bpiece[j] = q & 3;
    }

    const float vfloor = (PK != PK_ELU && p.pre_relu) ? 0.f : -__builtin_inff();
    float amax = 0.f;
    struct Stage {
        f32x4 a[AP];
        u32x4 bh[BPL], bl[BPL];
        f32x4 sc, sh;
        unsigned okmask;
        int cidx;
    };
    Stage st0, st1;                      // two register stages: loads run two K-steps ahead of the MFMAs
    int2 e_next = make_int2(0, (int)0x80000000);

    auto issue = [&](Stage& S, int s, bool prefetch) {
        f32x4(&areg)[AP] = S.a;
        u32x4(&bhreg)[BPL] = S.bh;
        u32x4(&blreg)[BPL] = S.bl;
        f32x4& sc = S.sc;
        f32x4& sh = S.sh;
        unsigned& okmask = S.okmask;
        int& st_cidx = S.cidx;
        const int2 e = e_next;
        if (prefetch) e_next = ktab[(s + 1) * 8 + kc];
        const int dy = (int)(short)(e.x & 0xffff);
        const int dx = e.x >> 16;
        const bool inval = e.y < 0;
        const int src = (e.y >> 30) & 1;
        const int coff = e.y & 0x3fffffff;
        const float* base = src ? p.src1 : p.src0;
        const int Cs = src ? p.Cs1 : p.Cs0;
        st_cidx = coff + (src ? p.C0 : 0);
        if (PK == PK_AFFINE && uni_b) {
            sc = *(const f32x4*)(p.pre_scale + aff_off + st_cidx);
            sh = *(const f32x4*)(p.pre_shift + aff_off + st_cidx);
        }
        const long tapoff = (long)(dy * p.W + dx) * Cs + coff;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            int iy = riy[i] + dy, ix = rix[i] + dx;
            bool ok = rb[i] >= 0 && !inval;
            long off;
            if (GEN) {
                if (p.pad_mode == FUSG_PAD_REFLECT) {
                    iy = iy < 0 ? -iy : (iy >= p.Hv ? 2 * p.Hv - 2 - iy : iy);
                    ix = ix < 0 ? -ix : (ix >= p.Wv ? 2 * p.Wv - 2 - ix : ix);
                } else if (p.pad_mode == FUSG_PAD_REPLICATE) {
                    iy = min(max(iy, 0), p.Hv - 1);
                    ix = min(max(ix, 0), p.Wv - 1);
                } else {
                    ok = ok && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
                }
                iy >>= p.ups; ix >>= p.ups;
                off = ((long)(rb[i] * p.H + iy) * p.W + ix) * Cs + coff;
            } else {
                ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                off = rowoff[i] * Cs + tapoff;
            }
            const float* ptr = base + off;
            if (PK == PK_AFFINE) { ptr = ok ? ptr : base; okmask |= (ok ? 1u : 0u) << i; }
            else ptr = ok ? ptr : p.zeros;                      // relu(0) = elu(0) = 0: no masking needed
            areg[i] = *(const f32x4*)ptr;
        }
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            if (BCH >= 256 * (j + 1) || t + 256 * j < BCH) {
                const long o = (long)brow[j] * p.K_pad + s * BK + bpiece[j] * 8;
                bhreg[j] = *(const u32x4*)(wh + o);
                blreg[j] = *(const u32x4*)(wl + o);
            }
        }
    };
    auto commit = [&](Stage& S, int buf) {
        f32x4(&areg)[AP] = S.a;
        u32x4(&bhreg)[BPL] = S.bh;
        u32x4(&blreg)[BPL] = S.bl;
        f32x4& sc = S.sc;
        f32x4& sh = S.sh;
        const unsigned okmask = S.okmask;
        const int st_cidx = S.cidx;
        // row r0 + 32 i has the same (row >> 2) & 3 for every i, so one swizzled offset serves all passes
        const int a_sw = ((((kc >> 1) ^ (r0 >> 2)) & 3) << 3) + ((kc & 1) << 2);
        _Float16* ah = Ah + buf * BM * LDH + r0 * LDH + a_sw;
        _Float16* al = Al + buf * BM * LDH + r0 * LDH + a_sw;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            f32x4 v = areg[i];
            if (PK == PK_ELU) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = elu1(v[c]);
            } else if (PK == PK_AFFINE) {
                if (!uni_b) {
                    const long o = (long)max(rb[i], 0) * p.pre_bstride + st_cidx;
                    sc = *(const f32x4*)(p.pre_scale + o);
                    sh = *(const f32x4*)(p.pre_shift + o);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaf(v[c], sc[c], sh[c]);
            }
            if (PK == PK_AFFINE) {
                const bool ok = (okmask >> i) & 1u;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = ok ? v[c] : 0.f;
            }
            h4 hi, lo;
            if constexpr (PK == PK_ELU) split4<false>(v, vfloor, hi, lo, amax); else split4(v, vfloor, hi, lo, amax);
            *(h4*)(ah + 32 * i * LDH) = hi;
            *(h4*)(al + 32 * i * LDH) = lo;
        }
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            if (BCH >= 256 * (j + 1) || t + 256 * j < BCH) {
                const int sw = ((bpiece[j] ^ (brow[j] >> 2)) & 3) << 3;
                *(u32x4*)(Bh + buf * BN * LDH + brow[j] * LDH + sw) = bhreg[j];
                *(u32x4*)(Bl + buf * BN * LDH + brow[j] * LDH + sw) = blreg[j];
            }
        }
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

    auto compute = [&](int buf) {
        const int a_off = (wm * TM * 32 + (lane & 31)) * LDH;
        const int b_off = (wn * TN * 32 + (lane & 31)) * LDH;
        const int rsw = (lane >> 2) & 3;             // (row >> 2) & 3 of this lane's rows
        const int hh = lane >> 5;                    // which 8 k of a 16-k chunk this lane feeds
        const _Float16* ahb = Ah + buf * BM * LDH + a_off;
        const _Float16* alb = Al + buf * BM * LDH + a_off;
        const _Float16* bhb = Bh + buf * BN * LDH + b_off;
        const _Float16* blb = Bl + buf * BN * LDH + b_off;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            h8 ah[TM], al[TM], bh[TN], bl[TN], bs[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *(const h8*)(ahb + i * 32 * LDH + (((2 * c + hh) ^ rsw) << 3));
                al[i] = *(const h8*)(alb + i * 32 * LDH + (((2 * c + hh) ^ rsw) << 3));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                bh[i] = *(const h8*)(bhb + i * 32 * LDH + (((2 * c + hh) ^ rsw) << 3));
                bl[i] = *(const h8*)(blb + i * 32 * LDH + (((2 * c + hh) ^ rsw) << 3));
                bs[i] = scale_m11(bh[i]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bs[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    // Software pipeline, distance 2: while step s is on the matrix cores, the loads of step s+2 are
    // in flight (stage registers) and step s+1 is being converted into the other LDS buffer.
    if (s_begin < s_end) {
        const int n = s_end - s_begin;
        e_next = ktab[s_begin * 8 + kc];
        issue(st0, s_begin, n > 1);
        if (n > 1) issue(st1, s_begin + 1, n > 2);
        commit(st0, 0);
        __syncthreads();
        int s = s_begin;
        // steady state: no conditionals inside, so that each half is one scheduling region in which the
        // staging VALU work (address generation, pre-op, fp16 split) is interleaved with the MFMAs
        // (in-order issue: a burst of back-to-back MFMAs would otherwise serialise the two pipes).
        for (; s + 4 < s_end; s += 2) {
            issue(st0, s + 2, true);
            compute(0);
            commit(st1, 1);
            FUSG_INTERLEAVE();
            __syncthreads();
            issue(st1, s + 3, true);
            compute(1);
            commit(st0, 0);
            FUSG_INTERLEAVE();
            __syncthreads();
        }
        for (; s + 1 < s_end; s += 2) {
            // even half: compute buf 0 (step s); st1 holds step s+1; refill st0 with step s+2
            if (s + 2 < s_end) issue(st0, s + 2, s + 3 < s_end);
            compute(0);
            commit(st1, 1);
            __syncthreads();
            // odd half: compute buf 1 (step s+1); st0 holds step s+2; refill st1 with step s+3
            if (s + 3 < s_end) issue(st1, s + 3, s + 4 < s_end);
            compute(1);
            if (s + 2 < s_end) commit(st0, 0);
            __syncthreads();
        }
        if (s < s_end) compute(0);             // odd count: the last step sits in buffer 0
    }

    report_range(p, amax);
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
        if (p.counters == nullptr) return;                                   // the host launches the reduce kernel
        if (!splitk_arrive(p, phase * gridDim.x + tile)) return;
        splitk_combine<BM, BN>(p, phase, mt, nt);
        return;
    }
    if (p.vec_epi) {
        __syncthreads();                                   // all waves are done with the staging buffers
        float* wlds = (float*)smem_h + wave * (TM * 32 * TN * 32);
        const int mwave = mt * BM + wm * TM * 32;
        epilogue_vec<TM, TN>(p, wlds, acc, lane, nt * BN + wn * TN * 32,
                             [&](int row, PixOff& po) { return pix_offsets(p, phase, mwave + row, po); },
                             [&](int i) -> float* {
                                 const int m0 = mwave + i * 32;
                                 if (m0 >= p.M) return nullptr;
                                 const int hw_ = p.Ho * p.Wo, b_ = m0 / hw_;
                                 return p.stats + ((long)b_ * p.stats_slots + (m0 - b_ * hw_) / 32) * p.Cout * 2;
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
            const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
            PixOff po;
            if (pix_offsets(p, phase, m, po)) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (nok[j]) epi_store(p, po, co[j], bias[j], wsc[j], acc[i][j][r]);
            }
        }
}

template <int TM, int TN, int WM, int WN>
hipError_t launch_h3(const ConvK& k, dim3 grid, hipStream_t s, int pk, bool gen) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const size_t lds = (size_t)2 * 2 * (BM + BN) * LDH * sizeof(_Float16);
    const void* fn = nullptr;
#define FUSG_PICK(PKV, GENV) fn = (const void*)conv_igemm_h3<TM, TN, WM, WN, PKV, GENV>
    if (pk == PK_NONE) { if (gen) FUSG_PICK(PK_NONE, true); else FUSG_PICK(PK_NONE, false); }
    else if (pk == PK_ELU) { if (gen) FUSG_PICK(PK_ELU, true); else FUSG_PICK(PK_ELU, false); }
    else { if (gen) FUSG_PICK(PK_AFFINE, true); else FUSG_PICK(PK_AFFINE, false); }
#undef FUSG_PICK
    if (hipError_t e = ensure_dyn_lds(fn, (int)lds); e != hipSuccess) return e;
    ConvK kk = k;
    void* args[] = {(void*)&kk};
    return hipLaunchKernel(fn, grid, dim3(256), args, lds, s);
}

}  // namespace fusg
