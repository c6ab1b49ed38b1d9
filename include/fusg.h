/*
 * fusg.h - C ABI of libfusg.so: the MI355X (gfx950) kernels behind the per-vehicle novel-view
 * synthesis hot path (stacked-hourglass -> Warp&Learn ICN -> VUnet -> EdgeConnect).
 *
 * The reference (alexj94/future_urban_scene_generation) has no FFI of its own: its hot path is
 * Python nn.Module code that delegates every op to PyTorch.  The drop-in boundary
 * is therefore the Python module surface (future_urban_scene_generation_amd/{stacked_hourglass,
 * warp_learn,vunet,edgeconnect}); this header is the thin C layer those modules bind with ctypes.
 * Each entry point names the reference op(s) it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes.  No torch types.  All pointers are DEVICE pointers
 *    borrowed from the caller (e.g. tensor.data_ptr()); the library never allocates or frees
 *    caller-visible memory and keeps no mutable state besides one-time per-device caches (kernel attributes, a 256-byte zero line) and the opt-in profiler.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Every call only enqueues
 *    work on that stream; nothing synchronises, so calls are graph-capturable.
 *  - Return 0 on success, negative fusg_status otherwise; fusg_last_error() gives a thread-local
 *    message.  Shapes are validated on the host before any launch.
 *  - Tensors are described by logical NCHW extents + element strides.  "NHWC-physical" means
 *    sc == 1 (channels contiguous), sw = Cs, sh = W*Cs, sn = H*W*Cs with Cs >= c, Cs % 4 == 0 and a
 *    16-byte aligned base: the layout every conv source must have.
 */
#ifndef FUSG_H
#define FUSG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FUSG_VERSION 115

typedef enum fusg_status {
    FUSG_OK = 0,
    FUSG_ERR_INVALID = -1,   /* bad shape / stride / enum            */
    FUSG_ERR_LAUNCH = -2,    /* HIP launch or runtime error          */
    FUSG_ERR_UNSUPPORTED = -3
} fusg_status;

typedef enum fusg_dtype { FUSG_F32 = 0, FUSG_U8 = 1, FUSG_I32 = 2 } fusg_dtype;

typedef struct fusg_tensor {
    void*   data;            /* device pointer, NULL = absent */
    int64_t n, c, h, w;      /* logical NCHW extents          */
    int64_t sn, sc, sh, sw;  /* strides in elements           */
    int32_t dtype;           /* fusg_dtype                    */
    int32_t _pad;
} fusg_tensor;

/* ---- convolution ------------------------------------------------------------------------- */

typedef enum fusg_pad_mode {
    FUSG_PAD_ZERO = 0, FUSG_PAD_REFLECT = 1,
    FUSG_PAD_REPLICATE = 2     /* clamp to the edge (the phase form of nn.Upsample(2) -> ReflectionPad2d(2) ->
                                  5x5 conv, pack.py: up2_phase_weights) */
} fusg_pad_mode;

/* op applied to every in-bounds source element while the im2col tile is staged (padding stays 0) */
typedef enum fusg_pre_op {
    FUSG_PRE_NONE = 0,
    FUSG_PRE_RELU = 1,
    FUSG_PRE_ELU = 2,          /* vunet/layers.py:12-15 (Activation) applied before the conv    */
    FUSG_PRE_AFFINE_RELU = 3,  /* relu(x*scale[b,c]+shift[b,c]): eval BatchNorm / InstanceNorm /
                                  custom LayerNorm followed by ReLU, normalise-on-load          */
    FUSG_PRE_AFFINE = 4
} fusg_pre_op;

typedef enum fusg_act {
    FUSG_ACT_NONE = 0, FUSG_ACT_RELU = 1, FUSG_ACT_TANH = 2, FUSG_ACT_SIGMOID = 3,
    FUSG_ACT_TANH01 = 4        /* (tanh(x)+1)/2, edgeconnect/networks.py:83 */
} fusg_act;

typedef enum fusg_store_mode {
    FUSG_STORE_NORMAL = 0,
    FUSG_STORE_D2S = 1,        /* DepthToSpace(2), DCR order, vunet/layers.py:173-196 */
    FUSG_STORE_S2D = 2         /* SpaceToDepth(2), vunet/layers.py:199-221            */
} fusg_store_mode;

/* Arithmetic of the conv contraction.
 * F32:   v_mfma_f32_32x32x2_f32, exact fp32 products and accumulation (157 TFLOP/s dense peak).
 * F16X3: operands split into fp16 pairs, a*w ~= ah*wh + ah*wl + al'*(wh*2^-11) on v_mfma_f32_16x16x32_f16 (halo kernel; 32x32x16 in
 *        the generic kernels) with fp32 accumulation; al' = (a - ah)*2^11 and the weights carry a per-output-channel power-of-two scale
 *        (wscale), so every operand in fp16's normal range [2^-14, 2^15) keeps >= 22 significant bits: the
 *        error against fp64 is at or below that of an fp32 fmaf chain for operand scales 1e-4 .. 1e4
 *        (tools/emu_split.py, tests/test_gpu_ops.py::test_f16x3_scale_sweep).  3 MFMA passes at the fp16 rate.
 *        Nothing is clamped: a launch that stages an operand with |x| >= 2^15 (or a non-finite one) sets
 *        *status = 1 and its output is unspecified; the caller must then redo the work with F32.
 * EMU_BF16 / EMU_BF16X2: evidence paths, not production: the F32 kernel with every staged activation rounded to 8 / 16
 *        significant bits (the caller passes weights rounded the same way in `wpack`): products of such operands are
 *        exact in fp32, so this is the arithmetic of a single-pass bf16 (resp. two-piece bf16) MFMA contraction with
 *        fp32 accumulation, at fp32-MFMA speed.  Used to decide BASELINE configs[4] with data.
 * BF16:  BASELINE configs[4]'s "bf16 MFMA conv path": launches that qualify for the halo kernel (the layers that carry the
 *        FLOPs) contract in ONE bf16 product per operand pair on v_mfma_f32_16x16x32_bf16 (operands rounded to bf16 as
 *        they are staged, weights from `wfrag_bf16`, fp32 accumulation); every other launch runs as F16X3, whose
 *        fields must be given too.  Measured: SSIM >= 0.9997 on every image output of the path, keypoint argmax NOT
 *        bit-exact - the hourglass module keeps F16X3 under this setting. */
typedef enum fusg_precision { FUSG_PREC_F32 = 0, FUSG_PREC_F16X3 = 1, FUSG_PREC_EMU_BF16 = 2, FUSG_PREC_EMU_BF16X2 = 3,
                              FUSG_PREC_BF16 = 4 } fusg_precision;

typedef enum fusg_tile {       /* workgroup tile (output pixels x output channels); 0 = auto */
    FUSG_TILE_AUTO = 0, FUSG_TILE_128x128 = 1, FUSG_TILE_128x64 = 2, FUSG_TILE_128x32 = 3,
    FUSG_TILE_64x64 = 4, FUSG_TILE_64x128 = 5
} fusg_tile;

/*
 * One fused convolution launch.  Replaces, per call site, the reference's chains of
 *   [torch.cat] -> [BatchNorm2d(eval)/InstanceNorm2d/LayerNorm -> ReLU | ELU] -> [ReflectionPad2d |
 *   ZeroPad2d | nn.Upsample(2)] -> nn.Conv2d / nn.ConvTranspose2d(k4,s2,p1) -> [+bias] ->
 *   [ReLU|tanh|sigmoid] -> [+residual] -> [DepthToSpace | SpaceToDepth]
 * (stacked_hourglass/models.py:22-42; warp_learn/models.py:84-90,106-110,157-159;
 *  vunet/layers.py:33-36,98-102,140-146; edgeconnect/networks.py:41-74,184-203).
 *
 * The GEMM view is  out[m, n] = sum_k A[m, k] * W[n, k],  m = (b, qy, qx) over the "q-space"
 * output grid, k = (tap, concat-channel).  `ktab` (built at weight-pack time, see
 * future_urban_scene_generation_amd/pack.py) has one int2 per 4 consecutive k:
 *     .x = (dy & 0xffff) | (dx << 16)       input offset of the tap (includes -pad and dilation)
 *     .y = channel offset in its source | src_select << 30 | invalid << 31
 * so  A[m, k] = pre_op(src[b, qy*stride + dy, qx*stride + dx, ch])  with padding handled by
 * `pad_mode` on the virtual (optionally 2x nearest-upsampled) input.  A transposed convolution
 * is `nphase` = 4 phase-convolutions with their own tables/weights and output offsets.
 */
typedef struct fusg_conv_desc {
    fusg_tensor src0;            /* NHWC-physical f32                                          */
    fusg_tensor src1;            /* optional second source (channel concat), same n,h,w        */
    fusg_tensor dst;             /* f32, any strides; logical extents of the FINAL output      */
    fusg_tensor res0, res1;      /* optional residuals, same logical extents as dst            */
    const float*   wpack;        /* [nphase][cout_pad][k_pad]                                  */
    const float*   bias;         /* [cout_pad]                                                 */
    const int32_t* ktab;         /* [nphase][k_pad/4][2]                                       */
    const float*   pre_scale;    /* [B or 1][c0k + c1k] for the AFFINE pre-ops                 */
    const float*   pre_shift;
    int64_t        pre_bstride;  /* elements between batches in pre_scale/shift (0 = shared)   */
    float*         workspace;    /* split-K partials: nphase*ksplit*M*cout_pad floats          */
    int32_t k_pad;               /* multiple of 32                                             */
    int32_t c0k;                 /* concat channels taken from src0 (multiple of 4)            */
    int32_t cout, cout_pad;      /* cout_pad multiple of 32                                    */
    int32_t stride;              /* q-space stride (1 or 2)                                    */
    int32_t upsample;            /* 0/1: sources are read through a 2x nearest upsample        */
    int32_t pad_mode;            /* fusg_pad_mode                                              */
    int32_t pre_op;              /* fusg_pre_op                                                */
    int32_t act;                 /* fusg_act                                                   */
    int32_t store_mode;          /* fusg_store_mode                                            */
    int32_t qh, qw;              /* q-space output grid                                        */
    int32_t nphase;              /* 1, or 4 for ConvTranspose2d(k4,s2,p1)                      */
    int32_t out_sy, out_sx;      /* dst y = qy*out_sy + out_oy[phase] (NORMAL store)           */
    int32_t out_oy[4], out_ox[4];
    int32_t dst_c_off;           /* write channels [dst_c_off, dst_c_off + cout') of dst       */
    int32_t tile;                /* fusg_tile                                                  */
    int32_t ksplit;              /* <=1: no split-K                                            */
    int32_t precision;           /* fusg_precision                                             */
    const void*    wpack_h;      /* F16X3 only: [nphase][2][cout_pad][k_pad] fp16 = (hi, lo) of w * s[n]   */
    /* Optional filter geometry (0 = unknown).  When given and the layer qualifies (F16X3, stride 1 - or stride 2
     * with wfrag_order 1 -, nphase 1, source channels % 32 == 0, qh % 8 == 0, qw % 16 == 0) the halo-tiled kernel
     * is used: taps must then be the dense kh x kw grid in (ky, kx) order with
     * dy = ky*dil - pad_h, dx = kx*dil - pad_w, which is what pack.py emits. */
    int32_t kh, kw, dil, pad_h, pad_w;
    int32_t wfrag_order;         /* 0: wfrag slabs in tap order.  1 (stride 2, k3/k4, pad 1, dil 1, one source with
                                    channels % 32 == 0, even H and W): slabs grouped by the parity quadrant of the input
                                    they read (pack.py: s2d_tap_order) - the launch then runs as four stride-1
                                    convolutions of the quarter-size parity sub-images on the halo kernel (odd
                                    H or W: generic gather).
                                    2 (one source with 4..24 channels, dil 1, k x k taps): wfrag holds the weights per
                                    16-k step of the tap-unit kernel (pack.py: frag_tapunit) - the whole few-channel
                                    halo is staged once and K runs over (tap, 8- or 4-channel unit)              */
    /* Halo kernel only: the (hi, lo) fp16 weights again, in MFMA-fragment order
     * [tap][chunk32][cout_pad/32][16-column half][hi|lo][64 lanes][8 halves], lane = (k >> 3) * 16 + column
     * (pack.py: frag_f16x3) = the B operand of v_mfma_f32_16x16x32_f16, so that a fragment is one contiguous
     * 1 KiB wave load.  NULL disables the halo kernel. */
    const void*    wfrag;
    /* Optional fused normalisation statistics of the conv OUTPUT (bias included): per image and per
     * 32-pixel slot (slot order is kernel-defined; finalisation is order-independent) the mean and the
     * centred sum of squares of every output channel: float [B][qh*qw/32][cout][2].  Requires
     * act == NONE, no residual, NORMAL store, nphase 1, ksplit <= 1, qh*qw % 32 == 0 and a
     * channel-contiguous 16-byte aligned dst with cout % 4 == 0 (else FUSG_ERR_UNSUPPORTED).
     * Feed to fusg_in_finalize_slots / fusg_ln_finalize_slots. */
    float*         stats_out;
    /* Optional (halo-kernel launches only, else FUSG_ERR_UNSUPPORTED): compute only these 8 x 16-pixel
     * patches of every image - indices into the row-major (qh/8) x (qw/16) patch grid, device memory. */
    const int32_t* tile_list;
    int32_t        tile_count;
    /* Origin of the computed qh x qw window in q-space (default 0, 0): output pixel (qy, qx) of the launch is
     * pixel (q_oy + qy, q_ox + qx) of the full convolution - lets a launch compute one edge row or column. */
    int32_t        q_oy, q_ox;
    /* Slots per image in stats_out (0 = qh*qw/32): lets several launches that each cover part of an image (the four
     * phase launches of a transposed convolution) write disjoint slot ranges of one buffer - offset the stats_out
     * pointer by first_slot*cout*2 floats. */
    int32_t        stats_slots;
    /* F16X3 only.  wscale[cout_pad] = 1 / s[n], s[n] the power of two the channel's weights were multiplied by
     * before the (hi, lo) split (pack.py: split_f16x3); the epilogue computes acc * wscale[n] + bias[n].
     * status: one device int32 (caller-owned, zeroed by the caller), set to 1 by the launch when an operand is
     * outside the split's range - see fusg_precision.  Sticky: the library never clears it. */
    const float*   wscale;
    int32_t*       status;
    /* FUSG_PREC_BF16 only: the weights rounded to bf16 (ties to even) in the halo kernel's fragment order
     * [tap][chunk32][cout_pad/32][16-column half][64 lanes][8 bf16], same tap order as `wfrag` (pack.py: frag_bf16).
     * NULL: the launch runs as F16X3.
     * With wfrag_order 2 (few-channel stems) it holds the tap-unit form instead: [k-step][cout_pad/32][64 lanes][8 bf16], the
     * `wfrag` layout of that order without the (hi | lo) axis (pack.py: frag_tapunit_bf16). */
    const void*    wfrag_bf16;
    /* FUSG_PREC_F32, optional: the fp32 weights in the halo kernel's fragment order for v_mfma_f32_16x16x4_f32,
     * [tap][chunk32][cout_pad/32][16-column half][h][64 lanes][4 floats] with lane = g * 16 + column holding
     * w[column][32 chunk + 16 h + 4 g + e], e = 0..3, same tap order as `wfrag` (pack.py: frag_f32).  Given (with the
     * filter geometry above), a qualifying FUSG_PREC_F32 launch runs on the halo kernel in exact fp32 instead of the
     * generic gather; NULL keeps the generic gather.
     * With wfrag_order 2 (few-channel stems) it holds the tap-unit form instead: [unit of 4 channels][cout_pad/32][64 lanes][2]
     * floats, lane = g * 32 + column holding w[column][4 unit + g], w[column][4 unit + 2 + g] (pack.py: frag_tapunit_f32). */
    const void*    wfrag_f32;
    /* Optional, split-K launches: `splitk_counters_len` device int32 words that are ZERO when the launch starts (the
     * launch leaves them zero).  Given, the workgroup that finishes a tile's last K-slice sums the slices (in slice
     * order: bit-reproducible) and applies the epilogue inside the same launch; NULL, a second kernel does. */
    int32_t*       splitk_counters;
    int32_t        splitk_counters_len;
    int32_t        _pad2;
} fusg_conv_desc;

int  fusg_conv2d(const fusg_conv_desc* d, void* stream);
/* Heuristic used when tile == AUTO / to size the workspace: fills tile, ksplit and returns the
 * workspace size in bytes (0 when ksplit <= 1). */
int64_t fusg_conv2d_plan(fusg_conv_desc* d);

/*
 * One pre-activation Bottleneck of the stacked hourglass in ONE launch (split-fp16 arithmetic, FUSG_PREC_F16X3):
 *   out = res + conv3_1x1( relu(bn3( conv2_3x3( relu(bn2( conv1_1x1( relu(bn1(x)) ))) ))) )
 * (stacked_hourglass/models.py:22-42; bn2 / bn3 are folded into conv1 / conv2 at pack time, bn1 is the per-channel
 * affine `pre_scale`, `pre_shift`; `res` is x itself or the block's 1x1 `downsample` conv of x, models.py:37-38).
 * planes P is 128 (every Bottleneck of the hourglass levels, `layer2`, `layer3` and `res`) or 64 (`layer1`): conv1
 * Cin -> P, conv2 P -> P, conv3 P -> 2 P; x has Cin % 32 == 0 channels, res and dst 2 P.
 * A workgroup owns an 8 x 8 pixel patch: conv1 is evaluated on the patch's 10 x 10 halo (zero outside the image, as
 * conv2's zero padding requires), its output and conv2's never leave LDS (already split into fp16 pairs), so the
 * block reads x and res once and writes out once - the three launches it replaces move 2.2x the bytes and, on the
 * hourglass's 4 x 4 .. 16 x 16 levels, are bound by launch-to-launch latency.
 * Weights: the `wfrag` copy (fusg_conv_desc.wfrag, wfrag_order 0) of each of the three convolutions, their `bias`
 * [cout_pad] and `wscale` [cout_pad] arrays.  Same range contract as F16X3 launches: *status = 1 when an operand
 * (bn1 output, or either intermediate) is outside the split's range; the caller then redoes the work in F32 with
 * three fusg_conv2d calls.  x, res, dst: NHWC-physical f32, same n, h, w.
 */
typedef struct fusg_bneck_desc {
    fusg_tensor x, res, dst;
    const float* pre_scale;      /* [Cin] bn1 as y = x * scale + shift, then ReLU               */
    const float* pre_shift;
    const void*  w1frag; const float* bias1; const float* wscale1;     /* conv1 (+bn2): Cin -> P, 1x1    */
    const void*  w2frag; const float* bias2; const float* wscale2;     /* conv2 (+bn3): P -> P, 3x3      */
    const void*  w3frag; const float* bias3; const float* wscale3;     /* conv3: P -> 2 P, 1x1           */
    int32_t*     status;
    int32_t      planes;         /* P: 64 or 128 */
    int32_t      exact_f32;      /* 0: split-fp16 (w*frag = the fusg_conv_desc.wfrag copies; wscale* and status required).
                                    1 (round 4): exact fp32 on v_mfma_f32_16x16x4_f32 - w*frag = the fusg_conv_desc.wfrag_f32
                                    copies (pack.py: frag_f32), wscale* and status unused */
} fusg_bneck_desc;
int fusg_hg_bottleneck(const fusg_bneck_desc* d, void* stream);

/* Load-time weight pre-packing on the HOST (no device work): everything fusg_conv_desc needs for one nn.Conv2d-style
 * filter weight[cout][cin][kh][kw] (torch layout, correlation form) whose input channels come from one source (c0 = cin)
 * or from two concatenated sources (the first c0 channels from src0: torch.cat([x, skip], 1) fused into the gather).
 * The Python modules do the same in future_urban_scene_generation_amd/pack.py (plus the reference-specific folds of
 * weight_norm / spectral_norm / eval BatchNorm, which are a few lines of arithmetic on the weights before this call:
 * vunet/layers.py:29-31, edgeconnect/networks.py:206-210, stacked_hourglass/models.py:11-16); the two are tested
 * against each other bit for bit.  Call fusg_pack_conv_sizes, allocate, call fusg_pack_conv_weights, upload. */
typedef struct fusg_pack_spec {
    int32_t cout, cin, kh, kw;
    int32_t c0;                  /* input channels taken from src0 (cin if there is one source)            */
    int32_t stride, pad, dil;    /* as nn.Conv2d; the taps are dy = ky*dil - pad, dx = kx*dil - pad        */
    int32_t upsample;            /* 1: the conv reads its sources through a 2x nearest upsample            */
    int32_t cin_pad;             /* K-channels of each source are padded to a multiple of this (4; 32 to make a
                                    few-channel layer eligible for the halo kernel)                        */
} fusg_pack_spec;
typedef struct fusg_pack_sizes {
    int32_t cout_pad, k_pad, c0k, c1k;   /* the fusg_conv_desc fields of the same names (c1k: K-channels of src1)   */
    int32_t wfrag_order;                 /* -1: no fragment-order copy (generic gather only), else fusg_conv_desc.wfrag_order */
    int32_t _pad;
    int64_t wpack_floats;                /* cout_pad * k_pad                                                        */
    int64_t ktab_ints;                   /* (k_pad / 4) * 2                                                         */
    int64_t wpack_h_halves;              /* 2 * cout_pad * k_pad                                                    */
    int64_t wfrag_halves;                /* 0 when wfrag_order < 0                                                  */
} fusg_pack_sizes;
int fusg_pack_conv_sizes(const fusg_pack_spec* spec, fusg_pack_sizes* sizes);
/* All pointers are HOST buffers of the sizes above (bias may be NULL = no bias; wfrag NULL = skip the copy);
 * bias_pad and wscale hold cout_pad floats; fp16 data is written as raw uint16 bit patterns. */
int fusg_pack_conv_weights(const fusg_pack_spec* spec, const float* weight, const float* bias, float* wpack, int32_t* ktab,
                           float* bias_pad, uint16_t* wpack_h, float* wscale, uint16_t* wfrag);

/* ---- normalisation statistics -------------------------------------------------------------- */

/* Per-(b, c) shifted sums over H*W in `nchunk` deterministic partials:
 * partial[b][chunk][c] = { sum(x - p), sum((x - p)^2) },  p = x[b, 0, 0, c].   x NHWC-physical. */
int fusg_chan_stats(const fusg_tensor* x, float* partial, int32_t nchunk, void* stream);
/* nn.InstanceNorm2d(affine=False, eps) statistics (warp_learn/models.py:56; edgeconnect/
 * networks.py:44): scale[b,c] = rstd, shift[b,c] = -mean*rstd (biased variance). */
int fusg_in_finalize(const fusg_tensor* x, const float* partial, int32_t nchunk, float eps,
                     float* scale, float* shift, void* stream);
/* Custom LayerNorm (warp_learn/models.py:26-35): per-sample mean and UNBIASED std over C*H*W,
 * eps added to std: scale[b,c] = gamma[c]/(std+eps), shift[b,c] = beta[c] - mean*scale[b,c]. */
int fusg_ln_finalize(const fusg_tensor* x, const float* partial, int32_t nchunk, float eps,
                     const float* gamma, const float* beta, float* scale, float* shift, void* stream);

/* The same two finalisations from the slot statistics a conv launch produced (fusg_conv_desc.stats_out):
 * slots [B][nslots][C][2] = (mean, centred M2) over 32 pixels each, combined in fp64 (Chan et al.). */
int fusg_in_finalize_slots(const float* slots, int32_t batch, int32_t nslots, int32_t channels, float eps,
                           float* scale, float* shift, void* stream);
int fusg_ln_finalize_slots(const float* slots, int32_t batch, int32_t nslots, int32_t channels, float eps,
                           const float* gamma, const float* beta, float* scale, float* shift, void* stream);

/* ---- elementwise / data movement ------------------------------------------------------------ */

/* dst = act(x*scale[b,c] + shift[b,c]) + res  (scale may be NULL: identity).  NHWC-physical.
 * InstanceNorm apply + residual of ResBlock / ResnetBlock (warp_learn/models.py:106-110;
 * edgeconnect/networks.py:198-199). */
int fusg_affine_act(const fusg_tensor* x, const float* scale, const float* shift, int64_t bstride,
                    int32_t act, const fusg_tensor* res, const fusg_tensor* dst, void* stream);
/* F.max_pool2d(x, 2, stride=2) (stacked_hourglass/models.py:72,149).  NHWC-physical. */
int fusg_maxpool2(const fusg_tensor* x, const fusg_tensor* dst, void* stream);
/* dst = up1 + nearest_upsample2(low) (stacked_hourglass/models.py:81-82).  NHWC-physical. */
int fusg_upsample2_add(const fusg_tensor* low, const fusg_tensor* up1, const fusg_tensor* dst, void* stream);
/* Generic strided copy dst[b,c,y,x] = src[b,c,y,x]; channels [src.c, dst_c_fill) of dst are
 * zero-filled (NCHW <-> NHWC conversion with channel padding at the module boundary). */
int fusg_copy4d(const fusg_tensor* src, const fusg_tensor* dst, int32_t dst_c_fill, void* stream);
/* dst = a + b with arbitrary strides (Sampler: mu + CPU-drawn noise, vunet/layers.py:166). */
int fusg_add4d(const fusg_tensor* a, const fusg_tensor* b, const fusg_tensor* dst, void* stream);
/* SpaceToDepth(2) / DepthToSpace(2), DCR order (vunet/layers.py:173-221). NHWC-physical; dst may
 * be a channel slice of a wider buffer. */
int fusg_space_to_depth2(const fusg_tensor* x, const fusg_tensor* dst, void* stream);
int fusg_depth_to_space2(const fusg_tensor* x, const fusg_tensor* dst, void* stream);
/* EdgeModel / InpaintingModel input assembly (edgeconnect/models.py:130-133, 236-238):
 * mode 0: dst = cat(images*(1-m)+m, edges*(1-m), m)      images [B,1,H,W] -> dst [B,3(+pad),H,W]
 * mode 1: dst = cat(images*(1-m)+m, edges)               images [B,3,H,W] -> dst [B,4,H,W]    */
int fusg_ec_inputs(const fusg_tensor* images, const fusg_tensor* edges, const fusg_tensor* masks,
                   const fusg_tensor* dst, int32_t mode, void* stream);
/* Second half of the "row-split" small-Cout convolution: a kh x kw conv with cout*kw <= 32 is run
 * as a kh x 1 implicit GEMM producing t[b, co*kw + kx, y, x] = sum_{ky,c} in[y+ky-pad, x, c]*w[co,c,ky,kx]
 * (kw x fewer MFMA flops than padding cout to 32), then
 *   dst[b, co, y, x] = act(bias[co] + sum_kx t[b, co*kw + kx, y, pad_x(x + kx - pad)])
 * with zero / reflect handling of the horizontal border.  t NHWC-physical, dst any strides.
 * Used for the 7x7 heads (warp_learn/models.py:182-183; edgeconnect/networks.py:72-73,123-124). */
int fusg_hshift_sum(const fusg_tensor* t, const float* bias, int32_t kw, int32_t pad, int32_t pad_mode,
                    int32_t act, const fusg_tensor* dst, void* stream);
/* Row-major first-occurrence argmax over H*W per (b, c) -> idx[b*C + c] = y*W + x (int32).
 * Integer contract of get_maxima (utils/keypoint_utils.py:85-88). */
int fusg_argmax_hw(const fusg_tensor* x, int32_t* idx, void* stream);
/* to_image(from_LAB=False) quantiser (warp_learn/planes_utils.py:111-114):
 * u8[b,y,x,c] = trunc(clip((x+1)/2*255, 0, 255)).  dst is a U8 tensor described as NCHW extents
 * with HWC strides. */
int fusg_to_image_u8(const fusg_tensor* x, const fusg_tensor* dst, void* stream);
/* merged = out*m + img*(1-m); u8 = trunc(merged*255) (trajectory_inference.py:126-129). */
int fusg_merge_u8(const fusg_tensor* out, const fusg_tensor* img, const fusg_tensor* mask,
                  const fusg_tensor* dst, void* stream);

/* ---- OpenCV-defined uint8 steps around the ICN (device versions; parity unpinned, see oracle/cv_host.py) ------- */
/* U8 images are fusg_tensors with dtype FUSG_U8, logical [n, 3, h, w] and HWC strides (sc = 1, sw >= 3).           */

/* cv2.warpPerspective(src, H, dsize=(w, h)) with INTER_LINEAR / BORDER_CONSTANT 0 (warp_learn/planes_utils.py:76-77)
 * for n images at once: minv = n row-major 3x3 DOUBLE matrices in DEVICE memory, each the INVERSE of that image's H
 * (OpenCV inverts H on the host too).  Coordinates in double, rounded to 1/32 pixel, 15-bit bilinear weights,
 * (sum + 2^14) >> 15.  src and dst may differ in h, w. */
int fusg_warp_perspective_u8(const fusg_tensor* src, const double* minv, const fusg_tensor* dst, void* stream);
/* The same for a list of jobs inside two image stacks (every plane of every vehicle of a frame in one launch,
 * pipeline.VehiclePipeline.run_frame): job k warps image index[2k] of src with minv[k] into image index[2k + 1] of dst;
 * index = DEVICE int32 [jobs][2]; images of dst no job names are left untouched (the caller zero-fills them:
 * planes_utils.py:57 starts from zeros). */
int fusg_warp_perspective_indexed_u8(const fusg_tensor* src, const double* minv, const int32_t* index, int32_t jobs,
                                     const fusg_tensor* dst, void* stream);
/* get_planes (warp_learn/planes_utils.py:11-37): dst[p] = frame * fillPoly(polygon p) for up to 8 polygons of up
 * to 8 int32 vertices.  pts_xy [nplanes][8][2] (x, y) and nverts [nplanes] are HOST arrays (read before return). */
int fusg_fill_poly_planes_u8(const fusg_tensor* frame, const int32_t* pts_xy, const int32_t* nverts, int32_t nplanes,
                             const fusg_tensor* dst, void* stream);
/* get_icn_inputs (warp_learn/models.py:323-366) for a batch of B vehicles: sketch [B] (RGB), central [B] (RGB, already
 * dst.h x dst.w), planes [B * P] (BGR, P <= 6, same frame size as sketch); geom = DEVICE int32 [B][8] =
 * (x0, y0, x1, y1, pad_x_before, pad_y_before, pad_x_after, pad_y_after) of square_crop_from_bbox
 * (utils/crop_utils.py:27-50).  Crop of the zero-padded frame -> cv2.resize INTER_LINEAR -> RGB/BGR2LAB (8-bit integer
 * path) -> (v/255 - 0.5)/0.5, written to dst f32 NHWC-physical [B, 3 * (P + 2), h, w] in the reference's channel
 * order (sketch, central, planes). */
int fusg_icn_inputs(const fusg_tensor* sketch, const fusg_tensor* central, const fusg_tensor* planes, const int32_t* geom,
                    const fusg_tensor* dst, void* stream);
/* cv2.cvtColor(x, COLOR_LAB2BGR) on uint8 (to_image(from_LAB=True), warp_learn/planes_utils.py:117). */
int fusg_lab2bgr_u8(const fusg_tensor* src, const fusg_tensor* dst, void* stream);
/* Resize-back + order-dependent masked paste (trajectory_inference.py:184-198) of V network images net [V] (u8,
 * e.g. 256 x 256) into ONE frame: a frame pixel covered by paste masks takes, from the LAST covering vehicle, that
 * vehicle's image resized (cv2.resize INTER_LINEAR) to its crop, padding removed, placed at crop_xy_min - or 0 where
 * the pixel lies outside that rectangle; masks u8 [V, 1, H, W] (non-zero = paste), geom as in fusg_icn_inputs. */
int fusg_paste_back_u8(const fusg_tensor* net, const fusg_tensor* masks, const int32_t* geom, const fusg_tensor* frame, void* stream);
/* The same with --inpaint (trajectory_inference.py:107-145): vehicle v first writes its inpainted box image rect[v] (u8,
 * EdgeConnect's merged output * 255, :126-129), resized (cv2.resize INTER_LINEAR, :130-131) to the rectangle rect_geom[v] =
 * (x0, y0, x1, y1, -, -, -, -) = bbox_new_img, unmasked into the running composite (:140-143), then pastes its masked
 * network image as above (:184-198); vehicles in index order, the last layer covering a pixel wins.  rect / rect_geom
 * (DEVICE int32 [V][8]) both NULL = fusg_paste_back_u8. */
int fusg_paste_layers_u8(const fusg_tensor* net, const fusg_tensor* masks, const int32_t* geom, const fusg_tensor* rect,
                         const int32_t* rect_geom, const fusg_tensor* frame, void* stream);

/* ---- frame-chain glue (pipeline.VehiclePipeline.run_frame): the uint8 -> float steps between the frame and the networks */
/* square_crop_from_bbox (utils/crop_utils.py:4-52) + cv2.resize INTER_LINEAR for V windows: src u8 HWC [1] (every
 * window cut from the same image: the frame, trajectory_inference.py:58-60) or [V] (one image per window: the central
 * crop of each vehicle's resized box, warp_learn/vehicle_utils.py:49-52); geom = DEVICE int32 [V][8] as in fusg_icn_inputs;
 * dst [V, 3, h, w].  mode 0: u8 HWC.  mode 1: f32 NHWC-physical, transforms.ToTensor + normalize: (v / 255 - mean[c]) /
 * std[c] (trajectory_inference.py:61-64; mean3 / std3 = HOST arrays of 3 floats, read before return).  mode 2: f32
 * NHWC-physical, to_tensor: v / 255 * 2 - 1 (utils/misc_utils.py:35-49).  A window of zero extent writes zeros. */
int fusg_crop_resize_u8(const fusg_tensor* src, const int32_t* geom, const fusg_tensor* dst, int32_t mode,
                        const float* mean3, const float* std3, void* stream);
/* The VUnet's first-frame inputs (trajectory_inference.py:203-228) for V vehicles: frame u8 HWC [1, 3, H, W]; masks u8
 * [V, 1, H, W] (non-zero = vehicle: the reference's ~src_sketch_mask); src_sketch / dst_sketch u8 HWC [V, 3, H, W];
 * geom = DEVICE int32 [V][8], the square window of each mask's bounding box (fusg_mask_bbox_geom).  Writes
 * x [V, 6, h, w] = cat[to_tensor(resize(crop(frame * mask))), to_tensor(resize(crop(src_sketch))[..., ::-1])] with
 * masked-frame pixels set to 255 where the resized src sketch is all zero, and y [V, 3, h, w] =
 * to_tensor(resize(crop(dst_sketch))[..., ::-1]); both f32 NHWC-physical. */
int fusg_vunet_inputs(const fusg_tensor* frame, const fusg_tensor* masks, const fusg_tensor* src_sketch,
                      const fusg_tensor* dst_sketch, const int32_t* geom, const fusg_tensor* x, const fusg_tensor* y, void* stream);
/* np.nonzero(mask) -> (x_min, y_min, x_max, y_max) (warp_learn/models.py:333-336; trajectory_inference.py:204-206) and
 * the square_crop_from_bbox geometry of that box (utils/crop_utils.py:13-50, Python's float arithmetic in double) for V
 * masks u8 [V, 1, H, W], entirely on the device: bbox = DEVICE int32 [V][4], geom = DEVICE int32 [V][8].  An empty mask
 * gives an all-zero geom row (the reference raises there and skips the vehicle). */
int fusg_mask_bbox_geom(const fusg_tensor* masks, int32_t* bbox, int32_t* geom, void* stream);
/* Heat-map argmax indices -> keypoints in frame pixels (utils/keypoint_utils.py:66-92 after F.interpolate to 256 =
 * (x0 / hm_w, y0 / hm_h); trajectory_inference.py:95-97: * crop side + crop_min - pad, in float64), stored float32
 * [V][nkp][2] - the pose fit's input.  idx = DEVICE int32 [V][nkp] (fusg_argmax_hw), geom as above (the detector
 * box's crop). */
int fusg_keypoints_to_frame(const int32_t* idx, const int32_t* geom, float* out, int32_t vehicles, int32_t nkp,
                            int32_t hm_w, int32_t hm_h, void* stream);

/* ---- pose fit ------------------------------------------------------------------------------- */
/*
 * The reference's pose fit ("CamPoseCalib": Levenberg-Marquardt on a Rodrigues vector + translation, utils/cpc.py:45-139
 * with the iteration / lambda policies of utils/pnp_utils.py:8-41), batched: one GPU thread per (vehicle, start).
 * utils/pnp_utils.py:43-115 (cpc_rodr_4_angles) runs it from four fixed start rotations per vehicle, each run ~52
 * iterations of 12 autograd backward passes on the host (5 s per vehicle measured); here all vehicles and starts of a
 * frame are one launch.  Reference behaviour kept: float32; only the first min(6, npoints) points enter the Jacobian
 * (cpc.py:30) while all points enter the cost and the returned error; every step is accepted; the loop ends when
 * iteration > max_iter (reference: 50).  Device arrays, row-major:
 *   points3d [B][npoints][3], points2d [B][npoints][2], focals [B][2], centers [B][2]   (npoints <= 16)
 *   rvec0 [nstarts][3], tvec0 [3]                                                       start parameters
 *   rvec, tvec [B][nstarts][3], err [B][nstarts]   (err = mean squared residual of the last evaluated iteration)
 * The choice of the best start and the sign flip of pnp_utils.py:117-130 are a few host operations on these outputs
 * (future_urban_scene_generation_amd/utils/pnp_utils.py).
 */
int fusg_pnp_cpc(const float* points3d, const float* points2d, const float* focals, const float* centers,
                 const float* rvec0, const float* tvec0, int32_t B, int32_t npoints, int32_t nstarts, int32_t max_iter,
                 float* rvec, float* tvec, float* err, void* stream);

/* ---- recorded passes ------------------------------------------------------------------------ */
/* A fusg_plan records the launch sequence of one pass (every fusg_* launch made by the recording thread between
 * fusg_plan_begin and fusg_plan_end, with its descriptors copied and its stream remembered; the calls also execute)
 * and replays it with ONE call - the batch-1 call pattern of the reference (trajectory_inference.py:55,65) is bound by
 * the interpreter's ~12 us per launch, not by the GPU.  Not a hipGraph: replay issues the same launches on the same
 * streams (overlap between the branches of a pass is kept; a captured graph serialises them on ROCm 7.2).
 * Every device pointer used while recording must stay valid, with the same meaning, until fusg_plan_destroy. */
typedef struct fusg_plan fusg_plan;
fusg_plan* fusg_plan_create(void);
void       fusg_plan_destroy(fusg_plan* p);
int        fusg_plan_begin(fusg_plan* p);         /* start recording on the calling thread */
int        fusg_plan_end(fusg_plan* p);
/* `waiter` (hipStream_t) may not pass this point before the work issued so far on `signaller` is done. */
int        fusg_plan_add_dependency(fusg_plan* p, void* waiter, void* signaller);
/* asynchronous copy of `bytes` from pinned host memory to the device on `stream`, repeated by every run.  `src` is a
 * ring of `nslots` buffers `slot_stride` bytes apart; run r reads slot r % nslots (the recording is run 0).  Before a
 * run, fusg_plan_next_slot() returns the slot that run will read and waits until the copies that last read it have
 * executed: fill that slot, then call fusg_plan_run. */
int        fusg_plan_add_h2d(fusg_plan* p, void* dst, const void* src, int64_t bytes, int32_t nslots, int64_t slot_stride, void* stream);
int        fusg_plan_next_slot(fusg_plan* p);
int64_t    fusg_plan_size(const fusg_plan* p);    /* recorded operations */
int        fusg_plan_run(fusg_plan* p);           /* re-issue the recording; asynchronous like the launches themselves */
/* The same replay issued by ONE HOST THREAD PER RECORDED STREAM (the caller's thread takes the stream of the first operation,
 * persistent worker threads of the library the others; a wait is held back until its event's record of this run has been issued).
 * For passes that one issuing thread bounds (small batches: ~300 operations of ~6.6 us of host time each).  Same launches, streams and
 * results as fusg_plan_run; returns when every stream's operations have been issued.  fusg_plan_streams: recorded streams (-1: none). */
int32_t    fusg_plan_streams(fusg_plan* p);
int        fusg_plan_run_mt(fusg_plan* p);
/* The same recording as ONE hipGraph (measurement path; the product replays plans).  fusg_plan_graph_capture
 * re-issues the recording under a thread-local stream capture begun on `capture_stream` (the stream the pass was recorded
 * from; the recorded dependencies fork and join the side streams) and instantiates the graph; its h2d copies read pinned
 * ring slot `slot` only.  The capture is always ended, also on failure; the error text names the recorded operation that
 * failed.  fusg_plan_graph_slot waits until the previous launch has consumed that slot and returns it; fill it, then
 * fusg_plan_graph_launch(p, stream).  fusg_plan_graph_nodes: nodes of the captured graph (-1: none). */
int        fusg_plan_graph_capture(fusg_plan* p, void* capture_stream, int32_t slot);
int64_t    fusg_plan_graph_nodes(const fusg_plan* p);
int        fusg_plan_graph_slot(fusg_plan* p);
int        fusg_plan_graph_launch(fusg_plan* p, void* stream);

/* ---- misc ----------------------------------------------------------------------------------- */

int         fusg_version(void);
const char* fusg_last_error(void);
/* Kernel family of this thread's last fusg_conv2d launch (tests assert that the intended path ran):
 * 0 generic fp32, 1 generic split-fp16, 2 halo, 3 halo in parity-quadrant form (stride 2), 4 tap-unit kernel
 * (few-channel stems), 5 halo kernel in single-pass bf16; -1 none yet. */
enum { FUSG_CONV_GENERIC_F32 = 0, FUSG_CONV_GENERIC_F16X3 = 1, FUSG_CONV_HALO = 2, FUSG_CONV_HALO_S2D = 3,
       FUSG_CONV_TAPUNIT = 4, FUSG_CONV_HALO_BF16 = 5, FUSG_CONV_BNECK = 6 /* fusg_hg_bottleneck */,
       FUSG_CONV_POINTWISE = 7 /* 1x1 from <= 8 channels: streaming fp32 FMA kernel, no matrix cores */,
       FUSG_CONV_SMALL = 8 /* small output images (<= 64 pixels): latency-built split-fp16 kernel (csrc/conv_kernel_small.h) */,
       FUSG_CONV_HALO_F32 = 9 /* halo kernel in exact fp32 (v_mfma_f32_16x16x4_f32, fusg_conv_desc.wfrag_f32) */,
       FUSG_CONV_TAPUNIT_F32 = 10 /* few-channel stems in exact fp32 (csrc/conv_kernel_tapunit_f32.h; wfrag_order 2 + wfrag_f32) */,
       FUSG_CONV_TAPUNIT_BF16 = 11 /* few-channel stems in single-pass bf16 (FUSG_PREC_BF16; wfrag_order 2 + wfrag_bf16) */ };
int         fusg_last_conv_kernel(void);
const char* fusg_arch(void);                      /* "gfx950" */
/* sizeof(fusg_tensor) / sizeof(fusg_conv_desc) as compiled, so that FFI bindings can verify their
 * struct mirrors. */
int         fusg_sizeof_tensor(void);
int         fusg_sizeof_conv_desc(void);
int         fusg_sizeof_bneck_desc(void);
/* Opt-in per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
 * kind 0 = conv implicit-GEMM kernel.  Disabled by default; enabling makes launches record two
 * events each.  fusg_prof_read synchronises the recorded events and returns totals since reset. */
void fusg_prof_enable(int on);
void fusg_prof_reset(void);
int  fusg_prof_read(int kind, double* total_ms, int64_t* launches, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* FUSG_H */
