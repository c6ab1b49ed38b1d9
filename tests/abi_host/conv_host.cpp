// A host WITHOUT Python or torch driving libfusg through include/fusg.h only: pack a torch-layout filter on the
// host (fusg_pack_conv_weights), upload, run one fused convolution launch (bias + ReLU) in split-fp16 and in exact-fp32 arithmetic,
// and check both against a double-precision loop.  Built and run by tests/test_gpu_abi_host.py (hipcc, links libfusg.so + the HIP runtime).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fusg.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define FUSGCHECK(x) do { int r_ = (x); if (r_ != FUSG_OK) { std::printf("fusg error %d (%s) at %s:%d\n", r_, fusg_last_error(), __FILE__, __LINE__); return 3; } } while (0)

template <class T> static T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    if (hipMalloc((void**)&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

static fusg_tensor nhwc(void* data, int n, int c, int h, int w) {
    fusg_tensor t;
    std::memset(&t, 0, sizeof t);
    t.data = data; t.n = n; t.c = c; t.h = h; t.w = w;
    t.sn = (int64_t)h * w * c; t.sh = (int64_t)w * c; t.sw = c; t.sc = 1;
    t.dtype = FUSG_F32;
    return t;
}

int main() {
    if (fusg_version() != FUSG_VERSION) { std::printf("header %d != library %d\n", FUSG_VERSION, fusg_version()); return 1; }
    const int B = 2, Cin = 64, Cout = 64, K = 3, H = 32, W = 32;
    std::vector<float> x((size_t)B * H * W * Cin), wt((size_t)Cout * Cin * K * K), bias(Cout);
    uint32_t s = 12345u;
    auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (float& v : x) v = rnd();
    for (float& v : wt) v = rnd() * 0.06f;
    for (float& v : bias) v = rnd() * 0.1f;

    // ---- host-side packing (what pack.py does for the Python modules)
    fusg_pack_spec spec;
    std::memset(&spec, 0, sizeof spec);
    spec.cout = Cout; spec.cin = Cin; spec.kh = K; spec.kw = K; spec.c0 = Cin; spec.stride = 1; spec.pad = 1; spec.dil = 1; spec.cin_pad = 4;
    fusg_pack_sizes sz;
    FUSGCHECK(fusg_pack_conv_sizes(&spec, &sz));
    std::vector<float> wpack(sz.wpack_floats), bias_pad(sz.cout_pad), wscale(sz.cout_pad);
    std::vector<int32_t> ktab(sz.ktab_ints);
    std::vector<uint16_t> wpack_h(sz.wpack_h_halves), wfrag(sz.wfrag_halves);
    FUSGCHECK(fusg_pack_conv_weights(&spec, wt.data(), bias.data(), wpack.data(), ktab.data(), bias_pad.data(), wpack_h.data(),
                                     wscale.data(), sz.wfrag_halves ? wfrag.data() : nullptr));
    if (sz.wfrag_order != 0) { std::printf("expected a fragment-order copy in tap order, got %d\n", sz.wfrag_order); return 1; }

    float *dx = upload(x), *dwp = upload(wpack), *dbias = upload(bias_pad), *dws = upload(wscale);
    int32_t* dkt = upload(ktab);
    uint16_t *dwh = upload(wpack_h), *dwf = upload(wfrag);
    float* dout = nullptr;
    int32_t* dstatus = nullptr;
    HIPCHECK(hipMalloc((void**)&dout, (size_t)B * H * W * Cout * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&dstatus, sizeof(int32_t)));
    HIPCHECK(hipMemset(dstatus, 0, sizeof(int32_t)));
    if (!dx || !dwp || !dbias || !dws || !dkt || !dwh || !dwf) { std::printf("upload failed\n"); return 2; }
    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));

    // ---- double-precision reference: relu(conv + bias), NHWC
    std::vector<double> ref((size_t)B * H * W * Cout);
    double refmax = 0.0;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int xx = 0; xx < W; ++xx)
                for (int n = 0; n < Cout; ++n) {
                    double acc = bias[n];
                    for (int ky = 0; ky < K; ++ky) {
                        const int iy = y + ky - 1;
                        if (iy < 0 || iy >= H) continue;
                        for (int kx = 0; kx < K; ++kx) {
                            const int ix = xx + kx - 1;
                            if (ix < 0 || ix >= W) continue;
                            const float* xp = &x[(((size_t)b * H + iy) * W + ix) * Cin];
                            const float* wp = &wt[(((size_t)n * Cin) * K + ky) * K + kx];
                            for (int c = 0; c < Cin; ++c) acc += (double)xp[c] * (double)wp[(size_t)c * K * K];
                        }
                    }
                    const double v = acc > 0.0 ? acc : 0.0;
                    ref[(((size_t)b * H + y) * W + xx) * Cout + n] = v;
                    if (v > refmax) refmax = v;
                }

    const int precisions[2] = {FUSG_PREC_F16X3, FUSG_PREC_F32};
    const char* names[2] = {"f16x3", "f32"};
    for (int pi = 0; pi < 2; ++pi) {
        fusg_conv_desc d;
        std::memset(&d, 0, sizeof d);
        d.src0 = nhwc(dx, B, Cin, H, W);
        d.dst = nhwc(dout, B, Cout, H, W);
        d.wpack = dwp; d.bias = dbias; d.ktab = dkt;
        d.k_pad = sz.k_pad; d.c0k = sz.c0k; d.cout = Cout; d.cout_pad = sz.cout_pad;
        d.stride = 1; d.pad_mode = FUSG_PAD_ZERO; d.pre_op = FUSG_PRE_NONE; d.act = FUSG_ACT_RELU; d.store_mode = FUSG_STORE_NORMAL;
        d.qh = H; d.qw = W; d.nphase = 1; d.out_sy = 1; d.out_sx = 1;
        d.precision = precisions[pi];
        d.wpack_h = dwh; d.wscale = dws; d.status = dstatus;
        d.kh = K; d.kw = K; d.dil = 1; d.pad_h = 1; d.pad_w = 1; d.wfrag_order = 0; d.wfrag = dwf;
        d.tile = FUSG_TILE_AUTO;
        d.ksplit = 1;                                                       // K whole (0 = let the planner split K on small grids: the generic gather)
        const int64_t ws_bytes = fusg_conv2d_plan(&d);                      // fills tile (and ksplit when it is 0)
        if (ws_bytes < 0) { std::printf("conv2d_plan: %s\n", fusg_last_error()); return 3; }
        float* ws = nullptr;
        if (ws_bytes > 0) { HIPCHECK(hipMalloc((void**)&ws, (size_t)ws_bytes)); d.workspace = ws; }
        HIPCHECK(hipMemsetAsync(dout, 0xff, (size_t)B * H * W * Cout * sizeof(float), stream));
        FUSGCHECK(fusg_conv2d(&d, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        const int kernel = fusg_last_conv_kernel();
        std::vector<float> out((size_t)B * H * W * Cout);
        HIPCHECK(hipMemcpy(out.data(), dout, out.size() * sizeof(float), hipMemcpyDeviceToHost));
        int32_t status = -1;
        HIPCHECK(hipMemcpy(&status, dstatus, sizeof status, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (size_t i = 0; i < out.size(); ++i) {
            const double e = std::fabs((double)out[i] - ref[i]);
            if (!(e <= worst)) worst = e;                                  // (a NaN ends up in `worst`)
        }
        std::printf("%s: kernel family %d, max abs error %.3e of max |out| %.3f (relative %.3e), range status %d\n", names[pi], kernel,
                    worst, refmax, worst / refmax, status);
        if (!(worst / refmax < 2e-6) || status != 0) { std::printf("ABI_HOST_FAILED\n"); return 1; }
        if (pi == 0 && kernel != 2) { std::printf("expected the halo kernel (family 2)\nABI_HOST_FAILED\n"); return 1; }
        if (ws) HIPCHECK(hipFree(ws));
    }
    std::printf("ABI_HOST_OK\n");
    return 0;
}
