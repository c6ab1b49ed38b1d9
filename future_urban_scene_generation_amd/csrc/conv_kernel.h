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
#pragma once
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
    int pre_relu;            // ReLU after the (optional) affine pre-op
    const _Float16* wpack_h; // f16x3 path: [nphase][2 (hi, lo)][cout_pad][k_pad] halves
    int vec_epi;             // destination / residuals allow 16-byte channel-contiguous epilogue accesses
    float* stats;            // optional fused norm statistics: [B][stats_slots][Cout][2] = (mean, M2) per 32-pixel slot
    int stats_slots;         // slots per image = qh*qw/32
    int qy0, qx0;            // origin of the computed Ho x Wo window in q-space (generic kernels; 0 for the halo kernel)
    const float* wscale;     // f16x3 path: per-output-channel 1/s of the power-of-two weight scale s applied at pack
                             // time (pack.split_f16x3); the epilogue computes acc * wscale[n] + bias[n].  NULL = 1.
    int* status;             // f16x3 path: set to 1 by any workgroup that stages an operand with |x| >= 2^15 (outside
                             // the range the fp16 split represents); the caller re-runs the pass in exact fp32
    int* counters;           // split-K: one arrival counter per (phase, tile), zero between launches (the last arriver
                             // resets it); NULL = the host launches conv_splitk_reduce instead
    int round_bits;          // exact-fp32 kernel only, 0 = off: round every staged activation to this many significant bits
                             // (8 = bf16, 16 = two bf16 pieces) - the numerics of a reduced-precision MFMA path emulated
                             // on the fp32 matrix cores (products of such operands are exact in fp32), for the bf16
                             // evidence of DESIGN.md / tests::test_reduced_precision_evidence; not a production path
    int touch_w;             // 1: a grid of at most TOUCH_MAX_WGS workgroups warms L2 with its weight ranges (l2_touch)
    const float* zeros;      // 256 B of zeros in device memory: where the gather of a zero-padded pixel reads.  It
                             // comes in through the kernel arguments so that the selected pointer stays a GLOBAL
                             // one (a select against the address of a __device__ variable degrades the load to
                             // flat_load, and every s_waitcnt after a flat load has to drain vmcnt AND lgkmcnt to 0)
};

enum { PK_NONE = 0, PK_ELU = 1, PK_AFFINE = 2 };   // compile-time pre-op kind

// L2 warm-up of a weight range ("touch"): an LDS-DMA load of ONE dword per 128-byte line into a 256-byte dummy region
// of LDS that nothing reads - no VGPR destination, so nothing has to stay live while the line travels.  Why: inside
// the pass every small layer meets its weights cold (each layer has its own, and ~35 GB of activations go through the
// caches between two uses), and the kernels fetch weight fragments only one or two K-steps ahead of their use - so a
// workgroup of a launch with few workgroups pays a memory round trip PER K-STEP.  Touching the workgroup's whole weight
// range once at kernel start turns ~36 dependent misses into one.  Measured on the pass (round 3, rocprofv3 kernel traces
// with and without it; the card of the "with" trace was ~5 % slower on large kernels): launches of <= 256 workgroups 7-15 %
// shorter (halo kernel at 16 x 16: 32.7 -> 29.9 us; fused Bottleneck at 4 x 4 .. 16 x 16: 43 -> 41 us); launches of 512+
// workgroups 5-12 % LONGER (whole pass, same card, FUSG_NO_TOUCH A/B with every grid touching: conv 25.1 -> 25.55 ms) (the lines are in L2 after
// the first workgroups, and the compiler drains every LDS-DMA before the first staged chunk is used), the generic
// gather 9-22 % longer at every size - so only small halo / Bottleneck grids do it (TOUCH_MAX_WGS).
typedef __attribute__((address_space(1))) const void fusg_gptr;
typedef __attribute__((address_space(3))) void fusg_lptr;
__device__ __forceinline__ void l2_touch(const void* g, void* lds_dummy) {
    __builtin_amdgcn_global_load_lds((fusg_gptr*)g, (fusg_lptr*)lds_dummy, 4, 0, 0);
}
constexpr int TOUCH_LDS_BYTES = 256;               // the dummy region: 64 lanes x 4 bytes
constexpr int TOUCH_MAX_WGS = 256;                 // grids up to this many workgroups warm L2 (see above)

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case FUSG_ACT_RELU: return fmaxf(v, 0.f);
        case FUSG_ACT_TANH: return tanhf(v);
        case FUSG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case FUSG_ACT_TANH01: return (tanhf(v) + 1.f) / 2.f;
        default: return v;
    }
}

// ELU(alpha=1) as ATen computes it on CPU: x > 0 ? x : exp(x) - 1 (v_exp_f32 based; abs error ~1e-7)
__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : __expf(v) - 1.f; }

// Pixel part of the destination mapping: offsets (in elements) of logical channel 0 of output
// pixel m in dst / res0 / res1.  Returns false for m >= M.
struct PixOff { long d, r0, r1; };
__device__ __forceinline__ bool pix_offsets(const ConvK& p, int phase, int m, PixOff& o) {
    if (m >= p.M) return false;
    const int hw = p.Ho * p.Wo;
    const int b = m / hw;
    const int rem = m - b * hw;
    const int oy = rem / p.Wo + p.qy0;
    const int ox = rem - (oy - p.qy0) * p.Wo + p.qx0;
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
__device__ __forceinline__ void epi_store(const ConvK& p, const PixOff& po, const PixOff& co, float bias, float wsc, float v) {
    v = act_apply(fmaf(v, wsc, bias), p.act);
    if (p.res0) v += p.res0[po.r0 + co.r0];
    if (p.res1) v += p.res1[po.r1 + co.r1];
    p.dst[po.d + co.d] = v;
}


// Vectorised epilogue.  The MFMA accumulator layout is column-per-lane (one output channel per lane,
// 16 pixels in registers): stored directly that is 64 four-byte stores (+64 residual loads) per lane.
// Instead the wave's tile takes a detour through LDS (wave-private region, no workgroup barrier) and
// comes back pixel-major, so each lane handles 4 consecutive channels of one pixel with 16-byte
// accesses: 4x fewer memory instructions, 256-byte contiguous runs per pixel.
// `pix(row, po)` maps a row of the wave tile to its destination pixel offsets (false = out of range).
// second half of the vectorised epilogue: the wave's tile is in `wlds` ([TM*32 rows][TN*32 columns] floats)
// Residual values of the wave's tile, fetched BEFORE the accumulators take their detour through LDS: the loads
// (16 bytes per lane and pass, TM * ITER passes) then travel while the tile is written to LDS, the waves meet and the
// rows are read back, instead of each pass waiting for its own load (the residual add cost a 128 -> 128 3x3 layer at
// 256 x 256 10 % of its time, tools/layer_variants_exp.py).  After the main loop the operand fragments are dead, so
// the 64 registers are there.
template <int TM, int TN> struct ResRegs {
    static constexpr int Q = TN * 8, R = 64 / Q, ITER = 32 / R;
    f32x4 v[TM][ITER];
};
template <int TM, int TN, typename PixFn>
__device__ __forceinline__ void res_prefetch(const ConvK& p, int lane, int ncol_base, PixFn pix, ResRegs<TM, TN>& rr) {
    constexpr int Q = ResRegs<TM, TN>::Q, R = ResRegs<TM, TN>::R, ITER = ResRegs<TM, TN>::ITER;
    const int c4 = lane % Q;
    const int n = ncol_base + c4 * 4;
    const bool nvalid = n < p.Cout;
    PixOff co;
    co.d = co.r0 = co.r1 = 0;
    if (nvalid) chan_offsets(p, n, co);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int row = i * 32 + it * R + lane / Q;
            PixOff po;
            const bool ok = pix(row, po);
            const float* ptr = (nvalid && ok) ? p.res0 + po.r0 + co.r0 : p.zeros;
            rr.v[i][it] = *(const f32x4*)ptr;
        }
}

template <int TM, int TN, typename PixFn, typename StatFn>
__device__ __forceinline__ void epilogue_rows(const ConvK& p, float* wlds, int lane, int ncol_base, PixFn pix, StatFn stat_base,
                                              const ResRegs<TM, TN>& rr, bool use_rr) {
    constexpr int PITCH = TN * 32;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int Q = TN * 8;                         // float4 groups per row
    constexpr int R = 64 / Q;                         // rows covered by one wave pass
    constexpr int ITER = 32 / R;                      // passes per 32-row MFMA tile
    const int c4 = lane % Q;
    const int n = ncol_base + c4 * 4;
    const bool nvalid = n < p.Cout;
    f32x4 bs = {0.f, 0.f, 0.f, 0.f}, wsc = {1.f, 1.f, 1.f, 1.f};
    PixOff co;
    co.d = co.r0 = co.r1 = 0;
    if (nvalid) {
        bs = *(const f32x4*)(p.bias + n);
        if (p.wscale) wsc = *(const f32x4*)(p.wscale + n);
        chan_offsets(p, n, co);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        f32x4 vals[ITER];
        bool full = true;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int row = i * 32 + it * R + lane / Q;
            PixOff po;
            const bool ok = pix(row, po);
            full = full && ok;
            f32x4 v = *(const f32x4*)(wlds + row * PITCH + c4 * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = act_apply(fmaf(v[c], wsc[c], bs[c]), p.act);
            vals[it] = v;
            if (nvalid && ok) {
                if (use_rr) v += rr.v[i][it];
                else if (p.res0) v += *(const f32x4*)(p.res0 + po.r0 + co.r0);
                if (p.res1) v += *(const f32x4*)(p.res1 + po.r1 + co.r1);
                *(f32x4*)(p.dst + po.d + co.d) = v;
            }
        }
        // Fused InstanceNorm / LayerNorm statistics: per (32-pixel slot, channel) mean and centred M2 of the
        // values just produced (two passes over registers, cross-lane sums by wavefront shuffles), combined
        // deterministically in fp64 by fusg_*_finalize_slots.  Only whole, in-range 32-row tiles take part.
        if (p.stats != nullptr) {
            float* sb = stat_base(i);
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < ITER; ++it) s1 += vals[it];
#pragma unroll
            for (int off = Q; off < 64; off <<= 1)
#pragma unroll
                for (int c = 0; c < 4; ++c) s1[c] += __shfl_xor(s1[c], off, 64);
            const f32x4 mean = s1 * (1.f / 32.f);
            f32x4 m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < ITER; ++it) { const f32x4 dv = vals[it] - mean; m2 += dv * dv; }
#pragma unroll
            for (int off = Q; off < 64; off <<= 1)
#pragma unroll
                for (int c = 0; c < 4; ++c) m2[c] += __shfl_xor(m2[c], off, 64);
            if (full && nvalid && lane < Q && sb != nullptr) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { sb[(n + c) * 2] = mean[c]; sb[(n + c) * 2 + 1] = m2[c]; }
            }
        }
    }
}

template <int TM, int TN, typename PixFn, typename StatFn>
__device__ __forceinline__ void epilogue_vec(const ConvK& p, float* wlds, const f32x16 (&acc)[TM][TN], int lane,
                                             int ncol_base, PixFn pix, StatFn stat_base) {
    constexpr int PITCH = TN * 32;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                wlds[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * PITCH + j * 32 + (lane & 31)] = acc[i][j][r];
    ResRegs<TM, TN> none;
    epilogue_rows<TM, TN>(p, wlds, lane, ncol_base, pix, stat_base, none, false);
}

// the same for accumulators of v_mfma_f32_16x16x32_f16 tiles: acc[row group of 16][column group of 16], C/D map
// col = lane & 15, row = (lane >> 4) * 4 + reg
template <int TM, int TN, typename PixFn, typename StatFn>
__device__ __forceinline__ void epilogue_vec16(const ConvK& p, float* wlds, const f32x4 (&acc)[2 * TM][2 * TN], int lane,
                                               int ncol_base, PixFn pix, StatFn stat_base, const ResRegs<TM, TN>& rr, bool use_rr) {
    constexpr int PITCH = TN * 32;
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                wlds[(i * 16 + (lane >> 4) * 4 + r) * PITCH + j * 16 + (lane & 15)] = acc[i][j][r];
    epilogue_rows<TM, TN>(p, wlds, lane, ncol_base, pix, stat_base, rr, use_rr);
}

// ---- in-launch split-K combine ---------------------------------------------------------------------------------
// Small-M layers split K over gridDim.z workgroups that each write a partial slab.  Instead of a second launch that
// sums the slabs (one more dependent kernel boundary per layer: ~8 us on a chain of 74 such layers per pass), the
// workgroup that arrives LAST at the tile's counter sums them - in slab order, so the result does not depend on the
// arrival order and is bit-identical to the separate reduce kernel's - and runs the epilogue.  Placement-independent
// hand-off as cdna_hip_programming.md §5 ("Projection GEMM at M = 256", item 2) prescribes: plain slab stores ->
// every wave s_waitcnt vmcnt(0) -> workgroup barrier -> lane 0: agent-scope release fence, vmcnt(0), relaxed agent-scope
// ticket; the last arriver: agent-scope acquire fence, vmcnt(0), barrier, plain loads.  Slabs are tiny (<= 64 KiB per
// tile and split), which is where the guide says the in-launch combine beats the extra boundary.
__device__ __forceinline__ bool splitk_arrive(const ConvK& p, int tile_id) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(p.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == p.ksplit - 1;
        if (last) {
            __hip_atomic_store(p.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}

template <int BM, int BN>
__device__ __forceinline__ void splitk_combine(const ConvK& p, int phase, int mt, int nt) {
    constexpr int N4 = BN / 4;
    for (int idx = threadIdx.x; idx < BM * N4; idx += 256) {
        const int m = mt * BM + idx / N4, c4 = nt * N4 + idx % N4;
        PixOff po;
        if (!pix_offsets(p, phase, m, po)) continue;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < p.ksplit; ++k)
            s += *(const f32x4*)(p.ws + ((long)(phase * p.ksplit + k) * p.M + m) * p.Cout_pad + c4 * 4);
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
}

// TM x TN 32x32 tiles per wave, WM x WN waves; PK = pre-op kind; GEN = generic addressing
// (reflect padding and/or fused 2x nearest upsample) vs the cheap zero-pad path.
template <int TM, int TN, int WM, int WN, int PK, bool GEN>
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
    const int hw = p.Ho * p.Wo;

    // per-thread staging rows
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
    // per-(b,c) affine parameters: one load per step when the whole tile lies in one sample
    bool uni_b = true;
    long aff_off = 0;
    if (PK == PK_AFFINE && p.pre_bstride != 0) {
        const int m_lo = mt * BM, m_hi = min(mt * BM + BM - 1, p.M - 1);
        uni_b = (m_lo / hw) == (m_hi / hw);
        aff_off = (long)(m_lo / hw) * p.pre_bstride;
    }
    const int2* ktab = p.ktab + (long)phase * (p.K_pad >> 2);
    const float* wrow = p.wpack + ((long)phase * p.Cout_pad + (long)nt * BN + r0) * p.K_pad + kc * 4;

    f32x4 areg[AP], breg[BP];
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    unsigned okmask = 0;
    int st_cidx = 0;

    // ---- stage 1: address generation + global loads (raw values stay in registers)
    int2 e_next = make_int2(0, (int)0x80000000);
    auto issue = [&](int s, bool prefetch) {
        const int2 e = e_next;                               // table entry fetched one step ahead
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
            off = ok ? off : 0;                              // branch-free: always a valid address
            areg[i] = *(const f32x4*)(base + off);
            okmask |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) breg[i] = *(const f32x4*)(wrow + (long)(32 * i) * p.K_pad + s * BK);
    };
    // ---- stage 2 (after the MFMAs of the current step): pre-op, zero the padding, write LDS
    auto commit = [&](int buf) {
        float* a = As + buf * BM * LDK + r0 * LDK + kc * 4;
        float* b = Bs + buf * BN * LDK + r0 * LDK + kc * 4;
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
            if (PK != PK_ELU && p.pre_relu) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
            }
            const bool ok = (okmask >> i) & 1u;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = ok ? v[c] : 0.f;
            if (p.round_bits) {                                  // round to nearest even at `round_bits` significant bits
                const int drop = 24 - p.round_bits;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float f = v[c];                        // (a scalar copy: __builtin_bit_cast of a vector ELEMENT
                    unsigned u = __builtin_bit_cast(unsigned, f);    //  reads element 0 on this compiler)
                    u = (u + ((1u << (drop - 1)) - 1u) + ((u >> drop) & 1u)) & ~((1u << drop) - 1u);
                    v[c] = __builtin_bit_cast(float, u);
                }
            }
            *(f32x4*)(a + 32 * i * LDK) = v;
        }
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
        e_next = ktab[s_begin * 8 + kc];
        issue(s_begin, s_begin + 1 < s_end);
        commit(0);
        __syncthreads();
        const int a_off = (wm * TM * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
        const int b_off = (wn * TN * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
        for (int s = s_begin; s < s_end; ++s) {
            const int buf = (s - s_begin) & 1;
            const bool more = s + 1 < s_end;
            if (more) issue(s + 1, s + 2 < s_end);
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
            if (more) commit(buf ^ 1);
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
        if (p.counters == nullptr) return;                                   // the host launches the reduce kernel
        if (!splitk_arrive(p, phase * gridDim.x + tile)) return;
        splitk_combine<BM, BN>(p, phase, mt, nt);
        return;
    }
    if (p.vec_epi) {
        __syncthreads();                                   // all waves are done with the staging buffers
        float* wlds = (float*)smem + wave * (TM * 32 * TN * 32);
        const int mwave = mt * BM + wm * TM * 32;
        epilogue_vec<TM, TN>(p, wlds, acc, lane, nt * BN + wn * TN * 32,
                             [&](int row, PixOff& po) { return pix_offsets(p, phase, mwave + row, po); },
                             [&](int i) -> float* {
                                 const int m0 = mwave + i * 32;            // slot = 32-row group inside its image
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

// one launcher per tile shape; instantiated in conv_tile_*.hip (separate TUs for parallel builds)
template <int TM, int TN, int WM, int WN>
hipError_t launch_tile(const ConvK& k, dim3 grid, hipStream_t s, int pk, bool gen) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
    const void* fn = nullptr;
#define FUSG_PICK(PKV, GENV) fn = (const void*)conv_igemm_f32<TM, TN, WM, WN, PKV, GENV>
    if (pk == PK_NONE) { if (gen) FUSG_PICK(PK_NONE, true); else FUSG_PICK(PK_NONE, false); }
    else if (pk == PK_ELU) { if (gen) FUSG_PICK(PK_ELU, true); else FUSG_PICK(PK_ELU, false); }
    else { if (gen) FUSG_PICK(PK_AFFINE, true); else FUSG_PICK(PK_AFFINE, false); }
#undef FUSG_PICK
    if (hipError_t e = ensure_dyn_lds(fn, (int)lds); e != hipSuccess) return e;
    ConvK kk = k;
    void* args[] = {(void*)&kk};
    return hipLaunchKernel(fn, grid, dim3(256), args, lds, s);
}

hipError_t launch_tile_128x128(const ConvK&, dim3, hipStream_t, int, bool);
hipError_t launch_tile_128x64(const ConvK&, dim3, hipStream_t, int, bool);
hipError_t launch_tile_128x32(const ConvK&, dim3, hipStream_t, int, bool);
hipError_t launch_tile_64x64(const ConvK&, dim3, hipStream_t, int, bool);
hipError_t launch_tile_64x128(const ConvK&, dim3, hipStream_t, int, bool);
hipError_t launch_h3_128x128(const ConvK&, dim3, hipStream_t, int, bool);
hipError_t launch_h3_128x64(const ConvK&, dim3, hipStream_t, int, bool);
hipError_t launch_h3_128x32(const ConvK&, dim3, hipStream_t, int, bool);
hipError_t launch_h3_64x64(const ConvK&, dim3, hipStream_t, int, bool);
hipError_t launch_h3_64x128(const ConvK&, dim3, hipStream_t, int, bool);

}  // namespace fusg
