// Device versions of the OpenCV-defined uint8 steps either side of the ICN (§8f-1, §8f-2):
// perspective warp of the texture planes, polygon plane masks, ICN input assembly (crop -> resize -> Lab -> normalise)
// and the way back (Lab -> BGR, resize-back + order-dependent masked paste).  The reference runs these on the host
// through opencv-python (warp_learn/planes_utils.py:11-118, warp_learn/models.py:323-366,
// trajectory_inference.py:184-198).  All of them are byte / integer work and HBM- or latency-bound: one thread per
// output pixel, coalesced 3-byte stores, gathers served by L2; no matrix cores.  The arithmetic is OpenCV's published
// 8-bit fixed-point algorithm for each op, restated independently in oracle/cv_host.py (parity unpinned: OpenCV is
// absent from the build container, see that file's header).
#include <math.h>
#include <mutex>
#include "common.h"

namespace fusg {

struct U8View { unsigned char* p; long sn, sh, sw; int n, h, w; };
static inline bool is_u8_hwc(const fusg_tensor& t, int c) {
    return t.data && t.dtype == FUSG_U8 && t.c == c && t.sc == 1 && t.sw >= c && t.sh >= t.w * t.sw && t.n >= 1 && t.h >= 1 && t.w >= 1 &&
           t.h < 32768 && t.w < 32768;
}
static inline U8View u8view(const fusg_tensor& t) {
    return U8View{(unsigned char*)t.data, t.sn, t.sh, t.sw, (int)t.n, (int)t.h, (int)t.w};
}
static inline unsigned blocks2d(long total) { return (unsigned)((total + 255) / 256); }

// ------------------------------------------------------------------------------------------------ warpPerspective
// dst(x, y) = bilinear(src, Minv * (x, y, 1)): coordinates in double, rounded to 1/32 pixel (INTER_BITS = 5), weights
// (32-ax)(32-ay)*32 ... in 15-bit fixed point (table entry (0,0) is (32767, 0, 0, 1): imgwarp.cpp initInterTab2D),
// (sum + 2^14) >> 15, constant border 0.
// index (optional, device int32 [jobs][2]): job n reads image index[2n] of src and writes image index[2n + 1] of dst.
__global__ __launch_bounds__(256) void warp_perspective_u8_kernel(U8View src, const double* __restrict__ minv, U8View dst, long total,
                                                                  const int* __restrict__ index) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % dst.w);
    const long r = idx / dst.w;
    const int y = (int)(r % dst.h);
    const int n = (int)(r / dst.h);
    const double* M = minv + (long)n * 9;
    const double X0 = M[0] * x + M[1] * y + M[2];
    const double Y0 = M[3] * x + M[4] * y + M[5];
    const double W0 = M[6] * x + M[7] * y + M[8];
    const double W = W0 != 0.0 ? 32.0 / W0 : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, X0 * W));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, Y0 * W));
    const long X = (long)rint(fX), Y = (long)rint(fY);                  // cvRound: to nearest, ties to even
    long sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const int ax = (int)(X & 31), ay = (int)(Y & 31);
    int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
    if ((ax | ay) == 0) { w00 = 32767; w11 = 1; }
    const unsigned char* s = src.p + (long)(index ? index[2 * n] : n) * src.sn;
    const bool y0ok = sy >= 0 && sy < src.h, y1ok = sy + 1 >= 0 && sy + 1 < src.h;
    const bool x0ok = sx >= 0 && sx < src.w, x1ok = sx + 1 >= 0 && sx + 1 < src.w;
    const unsigned char* p00 = s + sy * src.sh + sx * src.sw;
    unsigned char* d = dst.p + (long)(index ? index[2 * n + 1] : n) * dst.sn + (long)y * dst.sh + (long)x * dst.sw;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int v00 = (y0ok && x0ok) ? p00[c] : 0, v01 = (y0ok && x1ok) ? p00[src.sw + c] : 0;
        const int v10 = (y1ok && x0ok) ? p00[src.sh + c] : 0, v11 = (y1ok && x1ok) ? p00[src.sh + src.sw + c] : 0;
        const int acc = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
        const int o = (acc + (1 << 14)) >> 15;
        d[c] = (unsigned char)(o < 0 ? 0 : (o > 255 ? 255 : o));
    }
}

// ------------------------------------------------------------------------------------------------ fillPoly planes
// dst[p](y, x, :) = frame(y, x, :) * inside_or_on_outline(polygon p): cv::fillPoly at shift 0 = the 8-connected outline
// drawn left to right (LineIterator) + the even-odd scanline interior with 16.16 fixed-point edge crossings,
// x1 = ceil(crossing), x2 = floor(next crossing) (drawing.cpp CollectPolyEdges / FillEdgeCollection).
constexpr int MAXV = 8;
struct PolySet { int nv[8]; int px[8][MAXV]; int py[8][MAXV]; };

__device__ __forceinline__ bool on_line(int x, int y, int x0, int y0, int x1, int y1) {
    if (x1 < x0) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
    const int dx = x1 - x0, dyv = y1 - y0, dy = dyv < 0 ? -dyv : dyv, sy = dyv >= 0 ? 1 : -1;
    if (dx >= dy) {
        const int k = x - x0;
        if (k < 0 || k > dx) return false;
        const int yy = y0 + sy * (dx ? (int)((2L * dy * k + dx - 1) / (2L * dx)) : 0);
        return yy == y;
    }
    const int k = (y - y0) * sy;
    if (k < 0 || k > dy) return false;
    return x0 + (int)((2L * dx * k + dy - 1) / (2L * dy)) == x;
}

__global__ __launch_bounds__(256) void fill_poly_planes_kernel(U8View frame, PolySet ps, U8View dst, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % dst.w);
    const long r = idx / dst.w;
    const int y = (int)(r % dst.h);
    const int p = (int)(r / dst.h);
    const int nv = ps.nv[p];
    bool in = false;
    long xs[MAXV];
    int nx = 0;
    for (int i = 0; i < nv; ++i) {
        const int j = i == 0 ? nv - 1 : i - 1;
        const int ax = ps.px[p][j], ay = ps.py[p][j], bx = ps.px[p][i], by = ps.py[p][i];
        in = in || on_line(x, y, ax, ay, bx, by);
        if (ay == by) continue;
        const int y0 = ay < by ? ay : by, y1 = ay < by ? by : ay, xa = ay < by ? ax : bx;
        if (y < y0 || y >= y1) continue;
        const long num = ((long)bx - ax) << 16, den = (long)by - ay;
        const long dxf = num / den;                                    // C++ integer division truncates toward zero
        const long xc = ((long)xa << 16) + dxf * (y - y0);
        int k = nx++;
        while (k > 0 && xs[k - 1] > xc) { xs[k] = xs[k - 1]; --k; }   // insertion sort (<= 8 crossings)
        xs[k] = xc;
    }
    for (int k = 0; k + 1 < nx; k += 2) {
        const long x1 = (xs[k] + 65535) >> 16, x2 = xs[k + 1] >> 16;
        in = in || (x >= x1 && x <= x2);
    }
    const unsigned char* s = frame.p + (long)y * frame.sh + (long)x * frame.sw;
    unsigned char* d = dst.p + (long)p * dst.sn + (long)y * dst.sh + (long)x * dst.sw;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = in ? s[c] : 0;
}

// ------------------------------------------------------------------------------------------------ resize + Lab
// cv::resize INTER_LINEAR, uint8: source index / 11-bit weights of one destination coordinate (float arithmetic as in
// resize.cpp), horizontal pass in int, vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
__device__ __forceinline__ void resize_coef(int d, int ssize, int dsize, int& s, int& a0, int& a1) {
    const double scale = (double)ssize / dsize;
    float f = (float)((d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (si < 0) { f = 0.f; si = 0; }
    if (si >= ssize - 1) { f = 0.f; si = ssize - 1; }
    s = si;
    a0 = (int)rintf((1.f - f) * 2048.f);
    a1 = (int)rintf(f * 2048.f);
}

struct LabTabs { const unsigned short* gamma; const unsigned short* cbrt; int coef[9]; };

// one pixel RGB (or BGR) uint8 -> Lab uint8, OpenCV's integer path (color_lab.cpp RGB2Lab_b)
__device__ __forceinline__ void rgb2lab_px(const LabTabs& t, int r, int g, int b, int& L, int& A, int& B) {
    const int R = t.gamma[r], G = t.gamma[g], Bc = t.gamma[b];
    const int fX = t.cbrt[(R * t.coef[0] + G * t.coef[1] + Bc * t.coef[2] + (1 << 11)) >> 12];
    const int fY = t.cbrt[(R * t.coef[3] + G * t.coef[4] + Bc * t.coef[5] + (1 << 11)) >> 12];
    const int fZ = t.cbrt[(R * t.coef[6] + G * t.coef[7] + Bc * t.coef[8] + (1 << 11)) >> 12];
    const int Lscale = (116 * 255 + 50) / 100, Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    L = (Lscale * fY + Lshift + (1 << 14)) >> 15;
    A = (500 * (fX - fY) + 128 * (1 << 15) + (1 << 14)) >> 15;
    B = (200 * (fY - fZ) + 128 * (1 << 15) + (1 << 14)) >> 15;
    L = L < 0 ? 0 : (L > 255 ? 255 : L); A = A < 0 ? 0 : (A > 255 ? 255 : A); B = B < 0 ? 0 : (B > 255 ? 255 : B);
}

// per-device Lab tables: gamma[256], cbrt[3072] (ushort), built once on the host with OpenCV's formulas
static const unsigned short* lab_tables_dev(int coef[9]) {
    static std::mutex mu;
    static unsigned short* tabs[64] = {};
    static int coef_h[9];
    static bool coef_done = false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (!coef_done) {
        const double m[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
        const double wp[3] = {0.950456, 1.0, 1.088754};
        for (int i = 0; i < 9; ++i) coef_h[i] = (int)rint(4096.0 * m[i] / wp[i / 3]);
        coef_done = true;
    }
    for (int i = 0; i < 9; ++i) coef[i] = coef_h[i];
    if (!tabs[dev]) {
        static unsigned short host[256 + 3072];
        for (int i = 0; i < 256; ++i) {
            const float x = (float)i * (1.f / 255.f);
            const float v = x <= 0.04045f ? x * (1.f / 12.92f) : (float)pow(((double)x + 0.055) * (1. / 1.055), 2.4);
            const float q = rintf(255.f * 8.f * v);
            host[i] = (unsigned short)(q < 0.f ? 0.f : (q > 65535.f ? 65535.f : q));
        }
        for (int i = 0; i < 3072; ++i) {
            const float x = (float)i * (1.f / (255.f * 8.f));
            const float v = x < 0.008856f ? x * 7.787f + 0.13793103448275862f : (float)cbrt((double)x);   // (OpenCV: cvCbrt, a fast approximation)
            const float q = rintf(32768.f * v);
            host[256 + i] = (unsigned short)(q < 0.f ? 0.f : (q > 65535.f ? 65535.f : q));
        }
        unsigned short* q = nullptr;
        if (hipMalloc((void**)&q, sizeof(host)) != hipSuccess) return nullptr;
        if (hipMemcpy(q, host, sizeof(host), hipMemcpyHostToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(q); return nullptr; }
        tabs[dev] = q;
    }
    return tabs[dev];
}

// ICN input assembly for a batch of vehicles.  Per vehicle b: `nimg` images (sketch, then the planes) of the same
// frame size are cropped to the square window geom[b] = (x0, y0, x1, y1, pad_x_before, pad_y_before) of the
// zero-padded frame (utils/crop_utils.py:27-50), resized to out_h x out_w, converted to Lab and written as
// (lab/255 - 0.5)/0.5 into channels [3*slot(i), +3) of dst[b] (slot 0 = sketch, slot 1 = central crop, slots 2.. =
// planes).  The central crop (already out_h x out_w) skips crop and resize.
struct IcnIn {
    U8View sketch, central, planes;      // sketch [B], central [B], planes [B * nplanes]
    const int* geom;                     // [B][8]
    float* dst; long dsn, dsh, dsw;      // NHWC-physical f32
    int B, out_h, out_w, nplanes;
    LabTabs t;
};

__device__ __forceinline__ int crop_fetch(const unsigned char* img, const U8View& v, int px, int py, int c) {
    // pixel (px, py) of the un-padded frame, 0 in the padding
    return ((unsigned)px < (unsigned)v.w && (unsigned)py < (unsigned)v.h) ? img[(long)py * v.sh + (long)px * v.sw + c] : 0;
}

__global__ __launch_bounds__(256) void icn_inputs_kernel(IcnIn a, long total) {
    __shared__ unsigned short s_gamma[256];
    __shared__ unsigned short s_cbrt[3072];
    for (int i = threadIdx.x; i < 256; i += 256) s_gamma[i] = a.t.gamma[i];
    for (int i = threadIdx.x; i < 3072; i += 256) s_cbrt[i] = a.t.cbrt[i];
    __syncthreads();
    LabTabs t = a.t;
    t.gamma = s_gamma; t.cbrt = s_cbrt;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % a.out_w);
    long r = idx / a.out_w;
    const int y = (int)(r % a.out_h);
    r /= a.out_h;
    const int slot = (int)(r % (a.nplanes + 2));
    const int b = (int)(r / (a.nplanes + 2));
    int rgb[3];
    if (slot == 1) {
        const unsigned char* s = a.central.p + (long)b * a.central.sn + (long)y * a.central.sh + (long)x * a.central.sw;
        rgb[0] = s[0]; rgb[1] = s[1]; rgb[2] = s[2];
    } else {
        const U8View& v = slot == 0 ? a.sketch : a.planes;
        const unsigned char* img = v.p + (slot == 0 ? (long)b : (long)b * a.nplanes + (slot - 2)) * v.sn;
        const int* g = a.geom + b * 8;
        const int cw = g[2] - g[0], ch = g[3] - g[1];
        const int ox = g[0] - g[4], oy = g[1] - g[5];                  // crop origin in un-padded frame coordinates
        if (cw == a.out_w && ch == a.out_h) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] = crop_fetch(img, v, ox + x, oy + y, c);
        } else {
            int sx, ax0, ax1, sy, ay0, ay1;
            resize_coef(x, cw, a.out_w, sx, ax0, ax1);
            resize_coef(y, ch, a.out_h, sy, ay0, ay1);
            const int sx1 = sx + 1 < cw ? sx + 1 : cw - 1, sy1 = sy + 1 < ch ? sy + 1 : ch - 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int S0 = crop_fetch(img, v, ox + sx, oy + sy, c) * ax0 + crop_fetch(img, v, ox + sx1, oy + sy, c) * ax1;
                const int S1 = crop_fetch(img, v, ox + sx, oy + sy1, c) * ax0 + crop_fetch(img, v, ox + sx1, oy + sy1, c) * ax1;
                const int o = (((ay0 * (S0 >> 4)) >> 16) + ((ay1 * (S1 >> 4)) >> 16) + 2) >> 2;
                rgb[c] = o < 0 ? 0 : (o > 255 ? 255 : o);
            }
        }
    }
    int L, A, Bv;
    if (slot >= 2) rgb2lab_px(t, rgb[2], rgb[1], rgb[0], L, A, Bv);     // planes are BGR (planes_utils.py:88)
    else rgb2lab_px(t, rgb[0], rgb[1], rgb[2], L, A, Bv);               // sketch / central crop are RGB (models.py:355,358)
    float* d = a.dst + (long)b * a.dsn + (long)y * a.dsh + (long)x * a.dsw + slot * 3;
    d[0] = ((float)L / 255.f - 0.5f) / 0.5f;
    d[1] = ((float)A / 255.f - 0.5f) / 0.5f;
    d[2] = ((float)Bv / 255.f - 0.5f) / 0.5f;
}

// ------------------------------------------------------------------------------------------------ Lab -> BGR
__device__ __forceinline__ float lab_finv(float f) { return f <= (6.f / 29.f) ? (f - 16.f / 116.f) / 7.787f : f * f * f; }
__device__ __forceinline__ float srgb_gamma(float v) {
    return v <= 0.0031308f ? v * 12.92f : 1.055f * (float)pow((double)v, 1.0 / 2.4) - 0.055f;
}
__global__ __launch_bounds__(256) void lab2bgr_u8_kernel(U8View src, U8View dst, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % dst.w);
    const long r = idx / dst.w;
    const int y = (int)(r % dst.h);
    const int n = (int)(r / dst.h);
    const unsigned char* s = src.p + (long)n * src.sn + (long)y * src.sh + (long)x * src.sw;
    const float L = (float)s[0] * (100.f / 255.f), a = (float)s[1] - 128.f, b = (float)s[2] - 128.f;
    float fy = (L + 16.f) / 116.f, Y;
    if (L <= 903.3f * 0.008856f) { Y = L / 903.3f; fy = 7.787f * Y + 16.f / 116.f; } else Y = fy * fy * fy;
    const float X = lab_finv(fy + a / 500.f) * 0.950456f, Z = lab_finv(fy - b / 200.f) * 1.088754f;
    float R = 3.240479f * X + -1.53715f * Y + -0.498535f * Z;
    float G = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
    float Bl = 0.055648f * X + -0.204043f * Y + 1.057311f * Z;
    R = fminf(fmaxf(R, 0.f), 1.f); G = fminf(fmaxf(G, 0.f), 1.f); Bl = fminf(fmaxf(Bl, 0.f), 1.f);
    unsigned char* d = dst.p + (long)n * dst.sn + (long)y * dst.sh + (long)x * dst.sw;
    const float o[3] = {srgb_gamma(Bl) * 255.f, srgb_gamma(G) * 255.f, srgb_gamma(R) * 255.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float q = rintf(o[c]); d[c] = (unsigned char)(q < 0.f ? 0.f : (q > 255.f ? 255.f : q)); }
}

// ------------------------------------------------------------------------------------------------ paste back
// frame(y, x) <- for the LAST vehicle v (in index order) whose paste mask covers (y, x): the pixel of vehicle v's
// network image resized back to its crop (cv::resize INTER_LINEAR), with the crop padding removed and placed at
// crop_xy_min - or 0 where the pixel lies outside that rectangle (the reference pastes from a zero canvas,
// trajectory_inference.py:190-198); pixels no mask covers keep the frame's value.
// With `rect` (fusg_paste_layers_u8): every vehicle v brings TWO layers, in the reference's order with --inpaint
// (trajectory_inference.py:133-143 then :184-198, per vehicle, on the running composite): first its inpainted box image
// resized to the rectangle rgeom[v] = (x0, y0, x1, y1) and written there unmasked, then its masked network image; the
// last layer covering a pixel wins.
struct PasteIn { U8View net; U8View masks; U8View frame; const int* geom; int V; U8View rect; const int* rgeom; };   // geom [V][8] as in IcnIn

__device__ __forceinline__ void paste_resized_px(const unsigned char* img, const U8View& nv, int cx, int cy, int cw, int ch, unsigned char* d) {
    if (cw == nv.w && ch == nv.h) {
        const unsigned char* s = img + (long)cy * nv.sh + (long)cx * nv.sw;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        return;
    }
    int sx, ax0, ax1, sy, ay0, ay1;
    resize_coef(cx, nv.w, cw, sx, ax0, ax1);
    resize_coef(cy, nv.h, ch, sy, ay0, ay1);
    const int sx1 = sx + 1 < nv.w ? sx + 1 : nv.w - 1, sy1 = sy + 1 < nv.h ? sy + 1 : nv.h - 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int S0 = img[(long)sy * nv.sh + (long)sx * nv.sw + c] * ax0 + img[(long)sy * nv.sh + (long)sx1 * nv.sw + c] * ax1;
        const int S1 = img[(long)sy1 * nv.sh + (long)sx * nv.sw + c] * ax0 + img[(long)sy1 * nv.sh + (long)sx1 * nv.sw + c] * ax1;
        const int o = (((ay0 * (S0 >> 4)) >> 16) + ((ay1 * (S1 >> 4)) >> 16) + 2) >> 2;
        d[c] = (unsigned char)(o < 0 ? 0 : (o > 255 ? 255 : o));
    }
}
__global__ __launch_bounds__(256) void paste_back_kernel(PasteIn a, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % a.frame.w), y = (int)(idx / a.frame.w);
    unsigned char* d = a.frame.p + (long)y * a.frame.sh + (long)x * a.frame.sw;
    for (int v = a.V - 1; v >= 0; --v) {
        if (a.masks.p[(long)v * a.masks.sn + (long)y * a.masks.sh + (long)x * a.masks.sw]) {
            const int* g = a.geom + v * 8;
            const int cw = g[2] - g[0], ch = g[3] - g[1];
            // canvas rectangle: starts at crop_xy_min = (x0, y0); holds crop pixels [pad_before, size - pad_after)
            const int cx = x - g[0] + g[4], cy = y - g[1] + g[5];
            const int xa = g[6], ya = g[7];
            const bool inside = cx >= g[4] && cy >= g[5] && cx < cw - xa && cy < ch - ya;
            if (!inside) { d[0] = 0; d[1] = 0; d[2] = 0; return; }
            paste_resized_px(a.net.p + (long)v * a.net.sn, a.net, cx, cy, cw, ch, d);
            return;
        }
        if (a.rgeom) {
            const int* r = a.rgeom + v * 8;
            if (x >= r[0] && x < r[2] && y >= r[1] && y < r[3]) {
                paste_resized_px(a.rect.p + (long)v * a.rect.sn, a.rect, x - r[0], y - r[1], r[2] - r[0], r[3] - r[1], d);
                return;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------ frame-chain glue
// The steps of trajectory_inference.py:55-79, 200-228 between the uint8 frame and the networks' float inputs, so that a
// frame's vehicles go from detector boxes to rendered crops without leaving the device (pipeline.VehiclePipeline.run_frame).

// bilinear sample of a crop window (geom row g, as in IcnIn) of `img` resized to out_w x out_h: cv::resize INTER_LINEAR
template <class Fetch>
__device__ __forceinline__ void crop_resize_px(const int* g, int out_w, int out_h, int x, int y, Fetch fetch, int rgb[3]) {
    const int cw = g[2] - g[0], ch = g[3] - g[1];
    const int ox = g[0] - g[4], oy = g[1] - g[5];
    if (cw <= 0 || ch <= 0) { rgb[0] = rgb[1] = rgb[2] = 0; return; }
    if (cw == out_w && ch == out_h) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = fetch(ox + x, oy + y, c);
        return;
    }
    int sx, ax0, ax1, sy, ay0, ay1;
    resize_coef(x, cw, out_w, sx, ax0, ax1);
    resize_coef(y, ch, out_h, sy, ay0, ay1);
    const int sx1 = sx + 1 < cw ? sx + 1 : cw - 1, sy1 = sy + 1 < ch ? sy + 1 : ch - 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int S0 = fetch(ox + sx, oy + sy, c) * ax0 + fetch(ox + sx1, oy + sy, c) * ax1;
        const int S1 = fetch(ox + sx, oy + sy1, c) * ax0 + fetch(ox + sx1, oy + sy1, c) * ax1;
        const int o = (((ay0 * (S0 >> 4)) >> 16) + ((ay1 * (S1 >> 4)) >> 16) + 2) >> 2;
        rgb[c] = o < 0 ? 0 : (o > 255 ? 255 : o);
    }
}

// square_crop_from_bbox + cv2.resize for V windows (trajectory_inference.py:58-60; warp_learn/vehicle_utils.py:40-52),
// output uint8 HWC (mode 0), ToTensor + normalize (mode 1: (v/255 - mean) / std, trajectory_inference.py:61-64) or
// to_tensor (mode 2: v/255*2 - 1, utils/misc_utils.py:35-49) as f32 NHWC-physical.
struct CropIn { U8View src; const int* geom; U8View dst8; float* dstf; long dsn, dsh, dsw; int V, out_h, out_w, mode, src_per_v; float mean[3], stdv[3]; };
__global__ __launch_bounds__(256) void crop_resize_kernel(CropIn a, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % a.out_w);
    long r = idx / a.out_w;
    const int y = (int)(r % a.out_h);
    const int v = (int)(r / a.out_h);
    const unsigned char* img = a.src.p + (a.src_per_v ? (long)v * a.src.sn : 0);
    const U8View& sv = a.src;
    int rgb[3];
    crop_resize_px(a.geom + v * 8, a.out_w, a.out_h, x, y, [&](int px, int py, int c) { return crop_fetch(img, sv, px, py, c); }, rgb);
    if (a.mode == 0) {
        unsigned char* d = a.dst8.p + (long)v * a.dst8.sn + (long)y * a.dst8.sh + (long)x * a.dst8.sw;
        d[0] = (unsigned char)rgb[0]; d[1] = (unsigned char)rgb[1]; d[2] = (unsigned char)rgb[2];
        return;
    }
    float* d = a.dstf + (long)v * a.dsn + (long)y * a.dsh + (long)x * a.dsw;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float f = (float)rgb[c] / 255.f;
        d[c] = a.mode == 1 ? (f - a.mean[c]) / a.stdv[c] : f * 2.f - 1.f;
    }
}

// The VUnet's inputs (trajectory_inference.py:203-228): x = cat[to_tensor(masked frame crop), to_tensor(src sketch
// crop, channels reversed)], y_tilde = to_tensor(dst sketch crop, channels reversed); all three crops use the square
// window of the vehicle mask's bounding box, resized to out_w x out_h; masked-frame pixels whose (resized) src sketch
// is all zero become 255.
struct VuIn { U8View frame, mask, ssk, dsk; const int* geom; float* x; long xsn, xsh, xsw; float* y; long ysn, ysh, ysw; int V, out_h, out_w; };
__global__ __launch_bounds__(256) void vunet_inputs_kernel(VuIn a, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % a.out_w);
    long r = idx / a.out_w;
    const int y = (int)(r % a.out_h);
    const int v = (int)(r / a.out_h);
    const int* g = a.geom + v * 8;
    const unsigned char* mk = a.mask.p + (long)v * a.mask.sn;
    const unsigned char* s1 = a.ssk.p + (long)v * a.ssk.sn;
    const unsigned char* s2 = a.dsk.p + (long)v * a.dsk.sn;
    int mf[3], ss[3], ds[3];
    crop_resize_px(g, a.out_w, a.out_h, x, y, [&](int px, int py, int c) {
        if (!((unsigned)px < (unsigned)a.frame.w && (unsigned)py < (unsigned)a.frame.h)) return 0;
        return mk[(long)py * a.mask.sh + (long)px * a.mask.sw] ? (int)a.frame.p[(long)py * a.frame.sh + (long)px * a.frame.sw + c] : 0;
    }, mf);
    crop_resize_px(g, a.out_w, a.out_h, x, y, [&](int px, int py, int c) { return crop_fetch(s1, a.ssk, px, py, c); }, ss);
    crop_resize_px(g, a.out_w, a.out_h, x, y, [&](int px, int py, int c) { return crop_fetch(s2, a.dsk, px, py, c); }, ds);
    if ((ss[0] | ss[1] | ss[2]) == 0) mf[0] = mf[1] = mf[2] = 255;
    float* dx = a.x + (long)v * a.xsn + (long)y * a.xsh + (long)x * a.xsw;
    float* dy = a.y + (long)v * a.ysn + (long)y * a.ysh + (long)x * a.ysw;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dx[c] = (float)mf[c] / 255.f * 2.f - 1.f;
        dx[3 + c] = (float)ss[2 - c] / 255.f * 2.f - 1.f;
        dy[c] = (float)ds[2 - c] / 255.f * 2.f - 1.f;
    }
}

// Bounding box of the non-zero pixels of V masks (np.nonzero + min / max, warp_learn/models.py:333-336,
// trajectory_inference.py:204-206) and the square-crop geometry row of that box (utils/crop_utils.py:4-50), on the
// device: bbox [V][4] = (x_min, y_min, x_max, y_max), geom [V][8] as in IcnIn; an empty mask gives an all-zero geom
// row (cw = ch = 0: the consumers write zeros / paste nothing - the reference raises and skips the vehicle).
__global__ __launch_bounds__(256) void bbox_init_kernel(int* bbox, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) bbox[i] = (i & 2) ? -1 : 0x7fffffff;
}
// One workgroup = 16 rows of one vehicle's mask, 16 threads per row: each thread scans its row 16 bytes at a time (the 16
// threads of a row read 256 contiguous bytes per step), the workgroup reduces its extremes through wave shuffles and LDS and
// issues at most four atomics.  (Round 3's first form - one thread per 64-byte strip, four atomics per strip with a hit - took
// 117 us for eight 720 x 1280 masks: byte loads 64 bytes apart and ~25 thousand atomics on 32 addresses.)
constexpr int BBOX_ROWS = 16;
__global__ __launch_bounds__(256) void mask_bbox_kernel(U8View m, int* bbox, int bands) {
    const int v = blockIdx.x / bands, band = blockIdx.x - v * bands;
    const int y = band * BBOX_ROWS + (threadIdx.x >> 4), t = threadIdx.x & 15;
    int lo = 0x7fffffff, hi = -1;
    if (y < m.h) {
        const unsigned char* row = m.p + (long)v * m.sn + (long)y * m.sh;
        const bool vec = m.sw == 1 && ((uintptr_t)row & 15) == 0;
        for (int x0 = t * 16; x0 < m.w; x0 += 256) {
            if (vec && x0 + 16 <= m.w) {
                const uint4 q = *(const uint4*)(row + x0);
                const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (w4[k]) {
                        const int first = __builtin_ctz(w4[k]) >> 3, last = (31 - __builtin_clz(w4[k])) >> 3;
                        lo = min(lo, x0 + k * 4 + first);
                        hi = max(hi, x0 + k * 4 + last);
                    }
                }
            } else {
                const int x1 = min(m.w, x0 + 16);
                for (int x = x0; x < x1; ++x)
                    if (row[(long)x * m.sw]) { lo = min(lo, x); hi = max(hi, x); }
            }
        }
    }
    int ylo = hi >= 0 ? y : 0x7fffffff, yhi = hi >= 0 ? y : -1;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o));
        ylo = min(ylo, __shfl_xor(ylo, o)); yhi = max(yhi, __shfl_xor(yhi, o));
    }
    __shared__ int red[4][4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = lo; red[wave][1] = hi; red[wave][2] = ylo; red[wave][3] = yhi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = min(lo, red[w][0]); hi = max(hi, red[w][1]); ylo = min(ylo, red[w][2]); yhi = max(yhi, red[w][3]); }
        if (hi >= 0) {
            atomicMin(&bbox[v * 4 + 0], lo); atomicMax(&bbox[v * 4 + 2], hi);
            atomicMin(&bbox[v * 4 + 1], ylo); atomicMax(&bbox[v * 4 + 3], yhi);
        }
    }
}
__device__ __forceinline__ void square_axis(double c, double major, int size, int& lo, int& hi, int& pb, int& pa) {
    pb = 0; pa = 0;
    lo = (int)(c - major / 2.0);                                     // Python int(): truncation toward zero
    if (lo < 0) { pb = -lo; lo = 0; }
    hi = (int)(c + major / 2.0) + pb;
    if (hi > size) { pa = hi - size; hi = size + pa; }
}
__global__ __launch_bounds__(64) void bbox_geom_kernel(const int* bbox, int* geom, int V, int H, int W) {
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    const int x0 = bbox[v * 4], y0 = bbox[v * 4 + 1], x1 = bbox[v * 4 + 2], y1 = bbox[v * 4 + 3];
    int* g = geom + v * 8;
    if (x1 < 0) { for (int i = 0; i < 8; ++i) g[i] = 0; return; }
    const int sxd = x1 - x0, syd = y1 - y0;
    const double major = (double)max(sxd, syd) * 1.1;
    const double cx = (double)x0 + (double)sxd / 2.0, cy = (double)y0 + (double)syd / 2.0;
    int lx, hx, pxb, pxa, ly, hy, pyb, pya;
    square_axis(cx, major, W, lx, hx, pxb, pxa);
    square_axis(cy, major, H, ly, hy, pyb, pya);
    g[0] = lx; g[1] = ly; g[2] = hx; g[3] = hy; g[4] = pxb; g[5] = pyb; g[6] = pxa; g[7] = pya;
}

// Heat-map argmax -> keypoints in frame pixels (trajectory_inference.py:76-79, 95-97; utils/keypoint_utils.py:66-92):
// (x0 / hm_w) * crop_w + crop_x_min - pad_x in float64, stored as float32 (the pose fit's input type).
__global__ __launch_bounds__(256) void keypoints_to_frame_kernel(const int* idx, const int* geom, float* out, int total, int nkp, int hm_w, int hm_h) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int v = i / nkp;
    const int* g = geom + v * 8;
    const int k = idx[i];
    const double fx = (double)(k % hm_w) / (double)hm_w, fy = (double)(k / hm_w) / (double)hm_h;
    out[2 * i] = (float)(fx * (double)(g[2] - g[0]) + (double)g[0] - (double)g[4]);
    out[2 * i + 1] = (float)(fy * (double)(g[3] - g[1]) + (double)g[1] - (double)g[5]);
}

}  // namespace fusg

using namespace fusg;

static int warp_perspective_u8_impl(const fusg_tensor* src, const double* minv, const fusg_tensor* dst, const int32_t* index, int32_t jobs, void* stream) {
    FUSG_CHECK(src && dst && minv && is_u8_hwc(*src, 3) && is_u8_hwc(*dst, 3) && (index ? jobs >= 0 : src->n == dst->n),
               "warp_perspective_u8: u8 HWC tensors of 3 channels, same n (or an index of (source, destination) images per job)");
    FUSG_CHECK(src->data != dst->data, "warp_perspective_u8: in-place not supported");
    const long total = (index ? (long)jobs : dst->n) * dst->h * dst->w;
    if (total == 0) return FUSG_OK;
    hipLaunchKernelGGL(warp_perspective_u8_kernel, dim3(blocks2d(total)), dim3(256), 0, (hipStream_t)stream, u8view(*src), minv, u8view(*dst), total, index);
    FUSG_LAUNCH_CHECK("warp_perspective_u8");
    return FUSG_OK;
}
extern "C" int fusg_warp_perspective_u8(const fusg_tensor* src, const double* minv, const fusg_tensor* dst, void* stream) {
    return fusg::plan_dispatch(warp_perspective_u8_impl, stream, src, minv, dst, (const int32_t*)nullptr, (int32_t)0);
}
extern "C" int fusg_warp_perspective_indexed_u8(const fusg_tensor* src, const double* minv, const int32_t* index, int32_t jobs,
                                                const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(index != nullptr, "warp_perspective_indexed_u8: index is null");
    return fusg::plan_dispatch(warp_perspective_u8_impl, stream, src, minv, dst, index, jobs);
}

extern "C" int fusg_fill_poly_planes_u8(const fusg_tensor* frame, const int32_t* pts_xy, const int32_t* nverts, int32_t nplanes,
                                        const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(frame && dst && pts_xy && nverts && is_u8_hwc(*frame, 3) && is_u8_hwc(*dst, 3) && frame->n == 1 && dst->n == nplanes &&
               nplanes >= 1 && nplanes <= 8 && frame->h == dst->h && frame->w == dst->w, "fill_poly_planes_u8: shapes (<= 8 planes)");
    PolySet ps;
    memset(&ps, 0, sizeof(ps));
    for (int p = 0; p < nplanes; ++p) {                                  // HOST arrays: [nplanes][MAXV][2], [nplanes]
        FUSG_CHECK(nverts[p] >= 0 && nverts[p] <= MAXV, "fill_poly_planes_u8: %d vertices (max %d)", nverts[p], MAXV);
        ps.nv[p] = nverts[p];
        for (int i = 0; i < nverts[p]; ++i) { ps.px[p][i] = pts_xy[(p * MAXV + i) * 2]; ps.py[p][i] = pts_xy[(p * MAXV + i) * 2 + 1]; }
    }
    const long total = dst->n * dst->h * dst->w;
    hipLaunchKernelGGL(fill_poly_planes_kernel, dim3(blocks2d(total)), dim3(256), 0, (hipStream_t)stream, u8view(*frame), ps, u8view(*dst), total);
    FUSG_LAUNCH_CHECK("fill_poly_planes_u8");
    return FUSG_OK;
}

static int icn_inputs_impl(const fusg_tensor* sketch, const fusg_tensor* central, const fusg_tensor* planes, const int32_t* geom,
                               const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(sketch && central && planes && geom && dst && is_u8_hwc(*sketch, 3) && is_u8_hwc(*central, 3) && is_u8_hwc(*planes, 3),
               "icn_inputs: u8 HWC inputs");
    FUSG_CHECK(is_nhwc(*dst) && dst->n == sketch->n && central->n == sketch->n && planes->n % sketch->n == 0 && planes->n / sketch->n >= 1 &&
               planes->n / sketch->n <= 6 && planes->h == sketch->h && planes->w == sketch->w && central->h == dst->h && central->w == dst->w &&
               dst->c == 3 * (planes->n / sketch->n + 2), "icn_inputs: shapes");
    IcnIn a;
    a.sketch = u8view(*sketch); a.central = u8view(*central); a.planes = u8view(*planes);
    a.geom = geom; a.dst = (float*)dst->data; a.dsn = dst->sn; a.dsh = dst->sh; a.dsw = dst->sw;
    a.B = (int)sketch->n; a.out_h = (int)dst->h; a.out_w = (int)dst->w; a.nplanes = (int)(planes->n / sketch->n);
    const unsigned short* tabs = lab_tables_dev(a.t.coef);
    if (!tabs) { set_error("icn_inputs: cannot build the Lab tables"); return FUSG_ERR_LAUNCH; }
    a.t.gamma = tabs; a.t.cbrt = tabs + 256;
    const long total = (long)a.B * (a.nplanes + 2) * a.out_h * a.out_w;
    hipLaunchKernelGGL(icn_inputs_kernel, dim3(blocks2d(total)), dim3(256), 0, (hipStream_t)stream, a, total);
    FUSG_LAUNCH_CHECK("icn_inputs");
    return FUSG_OK;
}
extern "C" int fusg_icn_inputs(const fusg_tensor* sketch, const fusg_tensor* central, const fusg_tensor* planes, const int32_t* geom, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(icn_inputs_impl, stream, sketch, central, planes, geom, dst); }

static int lab2bgr_u8_impl(const fusg_tensor* src, const fusg_tensor* dst, void* stream) {
    FUSG_CHECK(src && dst && is_u8_hwc(*src, 3) && is_u8_hwc(*dst, 3) && src->n == dst->n && src->h == dst->h && src->w == dst->w, "lab2bgr_u8: shapes");
    const long total = dst->n * dst->h * dst->w;
    hipLaunchKernelGGL(lab2bgr_u8_kernel, dim3(blocks2d(total)), dim3(256), 0, (hipStream_t)stream, u8view(*src), u8view(*dst), total);
    FUSG_LAUNCH_CHECK("lab2bgr_u8");
    return FUSG_OK;
}
extern "C" int fusg_lab2bgr_u8(const fusg_tensor* src, const fusg_tensor* dst, void* stream) { return fusg::plan_dispatch(lab2bgr_u8_impl, stream, src, dst); }

static int paste_layers_u8_impl(const fusg_tensor* net, const fusg_tensor* masks, const int32_t* geom, const fusg_tensor* rect,
                                const int32_t* rect_geom, const fusg_tensor* frame, void* stream) {
    FUSG_CHECK(net && masks && geom && frame && is_u8_hwc(*net, 3) && is_u8_hwc(*frame, 3) && frame->n == 1 && masks->data &&
               masks->dtype == FUSG_U8 && masks->c == 1 && masks->n == net->n && masks->h == frame->h && masks->w == frame->w, "paste_back_u8: shapes");
    FUSG_CHECK((rect == nullptr) == (rect_geom == nullptr) && (!rect || (is_u8_hwc(*rect, 3) && rect->n == net->n)),
               "paste_layers_u8: the box images come with their rectangles, one per vehicle");
    PasteIn a;
    memset(&a, 0, sizeof(a));
    a.net = u8view(*net); a.masks = u8view(*masks); a.frame = u8view(*frame); a.geom = geom; a.V = (int)net->n;
    if (rect) { a.rect = u8view(*rect); a.rgeom = rect_geom; }
    const long total = frame->h * frame->w;
    hipLaunchKernelGGL(paste_back_kernel, dim3(blocks2d(total)), dim3(256), 0, (hipStream_t)stream, a, total);
    FUSG_LAUNCH_CHECK("paste_back_u8");
    return FUSG_OK;
}
extern "C" int fusg_paste_back_u8(const fusg_tensor* net, const fusg_tensor* masks, const int32_t* geom, const fusg_tensor* frame, void* stream) {
    return fusg::plan_dispatch(paste_layers_u8_impl, stream, net, masks, geom, (const fusg_tensor*)nullptr, (const int32_t*)nullptr, frame);
}
extern "C" int fusg_paste_layers_u8(const fusg_tensor* net, const fusg_tensor* masks, const int32_t* geom, const fusg_tensor* rect,
                                    const int32_t* rect_geom, const fusg_tensor* frame, void* stream) {
    return fusg::plan_dispatch(paste_layers_u8_impl, stream, net, masks, geom, rect, rect_geom, frame);
}

struct Norm3 { float m[3], s[3]; int has; };      // mode 1's mean / std, captured by value (host arrays at the ABI)
static int crop_resize_impl(const fusg_tensor* src, const int32_t* geom, const fusg_tensor* dst, int32_t mode, Norm3 nm, void* stream) {
    FUSG_CHECK(src && geom && dst && is_u8_hwc(*src, 3), "crop_resize_u8: u8 HWC source of 3 channels");
    FUSG_CHECK(mode >= 0 && mode <= 2 && (src->n == 1 || src->n == dst->n) && dst->n >= 1 && dst->h >= 1 && dst->w >= 1, "crop_resize_u8: mode / batch");
    CropIn a;
    memset(&a, 0, sizeof(a));
    a.src = u8view(*src); a.geom = geom; a.V = (int)dst->n; a.out_h = (int)dst->h; a.out_w = (int)dst->w; a.mode = mode;
    a.src_per_v = src->n == 1 ? 0 : 1;
    if (mode == 0) {
        FUSG_CHECK(is_u8_hwc(*dst, 3) && dst->data != src->data, "crop_resize_u8: mode 0 needs a u8 HWC destination (not in place)");
        a.dst8 = u8view(*dst);
    } else {
        FUSG_CHECK(is_nhwc(*dst) && dst->c == 3, "crop_resize_u8: modes 1 / 2 need an NHWC-physical f32 destination of 3 channels");
        a.dstf = (float*)dst->data; a.dsn = dst->sn; a.dsh = dst->sh; a.dsw = dst->sw;
        if (mode == 1) {
            FUSG_CHECK(nm.has, "crop_resize_u8: mode 1 needs mean / std (host arrays of 3)");
            for (int c = 0; c < 3; ++c) { a.mean[c] = nm.m[c]; a.stdv[c] = nm.s[c]; }
        }
    }
    const long total = (long)a.V * a.out_h * a.out_w;
    hipLaunchKernelGGL(crop_resize_kernel, dim3(blocks2d(total)), dim3(256), 0, (hipStream_t)stream, a, total);
    FUSG_LAUNCH_CHECK("crop_resize_u8");
    return FUSG_OK;
}
extern "C" int fusg_crop_resize_u8(const fusg_tensor* src, const int32_t* geom, const fusg_tensor* dst, int32_t mode, const float* mean3, const float* std3, void* stream) {
    Norm3 nm = {{0, 0, 0}, {1, 1, 1}, (mean3 && std3) ? 1 : 0};
    if (nm.has) for (int c = 0; c < 3; ++c) { nm.m[c] = mean3[c]; nm.s[c] = std3[c]; }
    return fusg::plan_dispatch(crop_resize_impl, stream, src, geom, dst, mode, nm);
}

static int vunet_inputs_impl(const fusg_tensor* frame, const fusg_tensor* masks, const fusg_tensor* src_sketch, const fusg_tensor* dst_sketch,
                             const int32_t* geom, const fusg_tensor* x, const fusg_tensor* y, void* stream) {
    FUSG_CHECK(frame && masks && src_sketch && dst_sketch && geom && x && y && is_u8_hwc(*frame, 3) && frame->n == 1 &&
               is_u8_hwc(*src_sketch, 3) && is_u8_hwc(*dst_sketch, 3), "vunet_inputs: u8 HWC frame [1] and sketches [V]");
    FUSG_CHECK(masks->data && masks->dtype == FUSG_U8 && masks->c == 1 && masks->n == src_sketch->n && masks->h == frame->h && masks->w == frame->w &&
               src_sketch->h == frame->h && src_sketch->w == frame->w && dst_sketch->n == masks->n && dst_sketch->h == frame->h && dst_sketch->w == frame->w,
               "vunet_inputs: masks [V, 1, H, W] u8 and sketches of the frame's size");
    FUSG_CHECK(is_nhwc(*x) && is_nhwc(*y) && x->c == 6 && y->c == 3 && x->n == masks->n && y->n == masks->n && x->h == y->h && x->w == y->w,
               "vunet_inputs: x [V, 6, h, w] and y [V, 3, h, w] NHWC-physical f32");
    VuIn a;
    a.frame = u8view(*frame); a.ssk = u8view(*src_sketch); a.dsk = u8view(*dst_sketch);
    a.mask = U8View{(unsigned char*)masks->data, masks->sn, masks->sh, masks->sw, (int)masks->n, (int)masks->h, (int)masks->w};
    a.geom = geom; a.x = (float*)x->data; a.xsn = x->sn; a.xsh = x->sh; a.xsw = x->sw;
    a.y = (float*)y->data; a.ysn = y->sn; a.ysh = y->sh; a.ysw = y->sw;
    a.V = (int)masks->n; a.out_h = (int)x->h; a.out_w = (int)x->w;
    const long total = (long)a.V * a.out_h * a.out_w;
    hipLaunchKernelGGL(vunet_inputs_kernel, dim3(blocks2d(total)), dim3(256), 0, (hipStream_t)stream, a, total);
    FUSG_LAUNCH_CHECK("vunet_inputs");
    return FUSG_OK;
}
extern "C" int fusg_vunet_inputs(const fusg_tensor* frame, const fusg_tensor* masks, const fusg_tensor* src_sketch, const fusg_tensor* dst_sketch,
                                 const int32_t* geom, const fusg_tensor* x, const fusg_tensor* y, void* stream) {
    return fusg::plan_dispatch(vunet_inputs_impl, stream, frame, masks, src_sketch, dst_sketch, geom, x, y);
}

static int mask_bbox_geom_impl(const fusg_tensor* masks, int32_t* bbox, int32_t* geom, void* stream) {
    FUSG_CHECK(masks && bbox && geom && masks->data && masks->dtype == FUSG_U8 && masks->c == 1 && masks->n >= 1 && masks->h >= 1 && masks->w >= 1 &&
               masks->h < 32768 && masks->w < 32768, "mask_bbox_geom: masks [V, 1, H, W] u8");
    const int V = (int)masks->n;
    U8View m{(unsigned char*)masks->data, masks->sn, masks->sh, masks->sw, V, (int)masks->h, (int)masks->w};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bbox_init_kernel, dim3((4 * V + 255) / 256), dim3(256), 0, s, bbox, 4 * V);
    const int bands = (m.h + BBOX_ROWS - 1) / BBOX_ROWS;
    hipLaunchKernelGGL(mask_bbox_kernel, dim3((unsigned)(V * bands)), dim3(256), 0, s, m, bbox, bands);
    hipLaunchKernelGGL(bbox_geom_kernel, dim3((V + 63) / 64), dim3(64), 0, s, (const int*)bbox, geom, V, m.h, m.w);
    FUSG_LAUNCH_CHECK("mask_bbox_geom");
    return FUSG_OK;
}
extern "C" int fusg_mask_bbox_geom(const fusg_tensor* masks, int32_t* bbox, int32_t* geom, void* stream) {
    return fusg::plan_dispatch(mask_bbox_geom_impl, stream, masks, bbox, geom);
}

static int keypoints_to_frame_impl(const int32_t* idx, const int32_t* geom, float* out, int32_t vehicles, int32_t nkp, int32_t hm_w, int32_t hm_h, void* stream) {
    FUSG_CHECK(idx && geom && out && vehicles >= 1 && nkp >= 1 && hm_w >= 1 && hm_h >= 1, "keypoints_to_frame: arguments");
    const int total = vehicles * nkp;
    hipLaunchKernelGGL(keypoints_to_frame_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, geom, out, total, nkp, hm_w, hm_h);
    FUSG_LAUNCH_CHECK("keypoints_to_frame");
    return FUSG_OK;
}
extern "C" int fusg_keypoints_to_frame(const int32_t* idx, const int32_t* geom, float* out, int32_t vehicles, int32_t nkp, int32_t hm_w, int32_t hm_h, void* stream) {
    return fusg::plan_dispatch(keypoints_to_frame_impl, stream, idx, geom, out, vehicles, nkp, hm_w, hm_h);
}
