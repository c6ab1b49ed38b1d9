// Batched pose fit of the reference's "CamPoseCalib" Levenberg-Marquardt (utils/cpc.py:45-139 steered by
// utils/pnp_utils.py:8-41), one GPU thread per (vehicle, start rotation): fusg_pnp_cpc (include/fusg.h).
//
// The reference builds the 12 x 6 Jacobian with 12 autograd backward passes per iteration (1.3 s per run of ~52
// iterations on the host, four runs per vehicle: 5 s per vehicle - measured in the build container).  The problem is
// 6 parameters x 12 points: here the Jacobian is analytic, a run is ~52 iterations of a few hundred float32
// operations in one thread, and every vehicle of a frame and all four starts run side by side in one launch.
// Reference behaviour kept (oracle/pnp.py spells it out): only the first min(6, n) points enter the Jacobian
// (cpc.py:30), every step is accepted, lambda follows check_lambda, the loop ends on iteration > max_iter, the error
// returned is that of the last evaluated body.  float32 throughout like the reference (lambda in double: a Python
// float there); compiled with -ffp-contract=off, so the arithmetic is the oracle's up to libm rounding.
#include "common.h"
#pragma clang diagnostic ignored "-Wpass-failed"     // the run-time-n instantiation cannot unroll its loops: expected

namespace fusg {

constexpr int PNP_MAXP = 16;
#define FUSG_UNROLL _Pragma("unroll")

struct PnpK {
    const float* p3; const float* p2; const float* focals; const float* centers; const float* rvec0; const float* tvec0;
    float* rvec; float* tvec; float* err;
    int B, n, S, max_iter;
};

__device__ static void rot_and_derivs(const float r[3], float R[9], float dR[3][9]) {
    const float th = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    const float u[3] = {r[0] / th, r[1] / th, r[2] / th};
    const float c = cosf(th), s = sinf(th), omc = 1.f - c;
    float uu[9], U[9] = {0.f, -u[2], u[1], u[2], 0.f, -u[0], -u[1], u[0], 0.f};
    FUSG_UNROLL
    for (int i = 0; i < 3; ++i)
        FUSG_UNROLL
        for (int j = 0; j < 3; ++j) uu[i * 3 + j] = u[i] * u[j];
    FUSG_UNROLL
    for (int i = 0; i < 3; ++i)
        FUSG_UNROLL
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = (i == j ? c : 0.f) + omc * uu[i * 3 + j] + U[i * 3 + j] * s;
    FUSG_UNROLL
    for (int k = 0; k < 3; ++k) {
        float du[3];
        FUSG_UNROLL
        for (int i = 0; i < 3; ++i) du[i] = ((i == k ? 1.f : 0.f) - u[i] * u[k]) / th;
        const float dU[9] = {0.f, -du[2], du[1], du[2], 0.f, -du[0], -du[1], du[0], 0.f};
        const float su = s * u[k], cu = c * u[k];
        FUSG_UNROLL
        for (int i = 0; i < 3; ++i)
            FUSG_UNROLL
            for (int j = 0; j < 3; ++j)
                dR[k][i * 3 + j] = (i == j ? -su : 0.f) + su * uu[i * 3 + j] + omc * (du[i] * u[j] + u[i] * du[j]) +
                                   cu * U[i * 3 + j] + s * dU[i * 3 + j];
    }
}

// inverse of a 6 x 6 matrix by Gauss-Jordan elimination with partial pivoting; false when a pivot is exactly 0
__device__ static bool inv6(const float A[36], float Ai[36]) {
    float M[6][12];
    FUSG_UNROLL
    for (int i = 0; i < 6; ++i)
        FUSG_UNROLL
        for (int j = 0; j < 6; ++j) { M[i][j] = A[i * 6 + j]; M[i][6 + j] = i == j ? 1.f : 0.f; }
    FUSG_UNROLL
    for (int c = 0; c < 6; ++c) {
        int p = c;
        float best = fabsf(M[c][c]);
        FUSG_UNROLL
        for (int i = c + 1; i < 6; ++i)
            if (fabsf(M[i][c]) > best) { best = fabsf(M[i][c]); p = i; }
        if (!(best > 0.f)) return false;                          // zero or NaN pivot
        FUSG_UNROLL
        for (int i = c + 1; i < 6; ++i)                             // row swap by selects: no dynamic register index
            FUSG_UNROLL
            for (int j = 0; j < 12; ++j) {
                const float a = M[c][j], b = M[i][j];
                M[c][j] = p == i ? b : a;
                M[i][j] = p == i ? a : b;
            }
        const float d = 1.f / M[c][c];
        FUSG_UNROLL
        for (int j = 0; j < 12; ++j) M[c][j] *= d;
        FUSG_UNROLL
        for (int i = 0; i < 6; ++i) {
            if (i == c) continue;
            const float f = M[i][c];
            FUSG_UNROLL
            for (int j = 0; j < 12; ++j) M[i][j] -= f * M[c][j];
        }
    }
    FUSG_UNROLL
    for (int i = 0; i < 6; ++i)
        FUSG_UNROLL
        for (int j = 0; j < 6; ++j) Ai[i * 6 + j] = M[i][6 + j];
    return true;
}

// NT > 0: the point count as a compile-time constant - every loop below unrolls and every array (J, err, the 6 x 12
// elimination tableau ...) lives in registers; the arithmetic and its order are those of the NT = 0 (run-time n) form,
// whose arrays sit in scratch memory and which took 3.1 ms for a frame's 32 runs against 0.2 ms for this one.
template <int NT>
__global__ __launch_bounds__(64) void pnp_cpc_kernel(const PnpK k) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= k.B * k.S) return;
    const int b = idx / k.S, st = idx - b * k.S;
    const int n = NT > 0 ? NT : k.n, nj = n < 6 ? n : 6;
    constexpr int NP = NT > 0 ? NT : PNP_MAXP;
    float P[NP][3], q[NP][2];
    FUSG_UNROLL
    for (int i = 0; i < n; ++i) {
        FUSG_UNROLL
        for (int j = 0; j < 3; ++j) P[i][j] = k.p3[((long)b * n + i) * 3 + j];
        FUSG_UNROLL
        for (int j = 0; j < 2; ++j) q[i][j] = k.p2[((long)b * n + i) * 2 + j];
    }
    const float fx = k.focals[b * 2], fy = k.focals[b * 2 + 1], cx = k.centers[b * 2], cy = k.centers[b * 2 + 1];
    float prm[6] = {k.rvec0[st * 3], k.rvec0[st * 3 + 1], k.rvec0[st * 3 + 2], k.tvec0[0], k.tvec0[1], k.tvec0[2]};
    float J[12][6], err[2 * NP], prev[2 * NP], upd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    FUSG_UNROLL
    for (int i = 0; i < 12; ++i)
        FUSG_UNROLL
        for (int j = 0; j < 6; ++j) J[i][j] = 0.f;
    FUSG_UNROLL
    for (int i = 0; i < 2 * NP; ++i) { err[i] = 0.f; prev[i] = 0.f; }
    double lam = 0.0, factor = 2.0;
    bool have_lam = false;
    int nerr = 0;                                                     // 0: no body yet, 1: cur only, 2: prev and cur
    int it = 0;
    for (;;) {
        if (nerr > 0) {                                               // check_iteration (pnp_utils.py:8-24)
            float gmax = 0.f, un = 0.f;
            double pn = 0.0;
            FUSG_UNROLL
            for (int j = 0; j < 6; ++j) {
                float g = 0.f;
                FUSG_UNROLL
                for (int i = 0; i < 2 * nj; ++i) g += J[i][j] * err[i];
                gmax = fmaxf(gmax, fabsf(g));
                un += upd[j] * upd[j];
                const double pp = (double)prm[j] - (double)upd[j];
                pn += pp * pp;
            }
            if (gmax < 1e-8f) break;
            if ((double)sqrtf(un) < 1e-8 * (sqrt(pn) + 1e-8)) break;
            if (it > k.max_iter) break;
        }
        float R[9], dR[3][9];
        rot_and_derivs(prm, R, dR);
        FUSG_UNROLL
        for (int i = 0; i < 2 * n; ++i) prev[i] = err[i];             // becomes prev_error once this body's error is cur_error
        float pc[NP][3];
        FUSG_UNROLL
        for (int i = 0; i < n; ++i) {
            FUSG_UNROLL
            for (int a = 0; a < 3; ++a) pc[i][a] = prm[3 + a] + (R[a * 3] * P[i][0] + R[a * 3 + 1] * P[i][1] + R[a * 3 + 2] * P[i][2]);
            const float iz = 1.f / pc[i][2];
            err[2 * i] = (fx * pc[i][0] * iz + cx) - q[i][0];
            err[2 * i + 1] = (fy * pc[i][1] * iz + cy) - q[i][1];
        }
        FUSG_UNROLL
        for (int i = 0; i < nj; ++i) {
            const float x = pc[i][0], y = pc[i][1], z = pc[i][2];
            const float dpx[3] = {fx / z, 0.f, -fx * x / (z * z)}, dpy[3] = {0.f, fy / z, -fy * y / (z * z)};
            FUSG_UNROLL
            for (int kk = 0; kk < 3; ++kk) {
                float dk[3];
                FUSG_UNROLL
                for (int a = 0; a < 3; ++a) dk[a] = dR[kk][a * 3] * P[i][0] + dR[kk][a * 3 + 1] * P[i][1] + dR[kk][a * 3 + 2] * P[i][2];
                J[2 * i][kk] = dpx[0] * dk[0] + dpx[1] * dk[1] + dpx[2] * dk[2];
                J[2 * i + 1][kk] = dpy[0] * dk[0] + dpy[1] * dk[1] + dpy[2] * dk[2];
            }
            FUSG_UNROLL
            for (int a = 0; a < 3; ++a) { J[2 * i][3 + a] = dpx[a]; J[2 * i + 1][3 + a] = dpy[a]; }
        }
        float A[36], sum = 0.f, dmax = 0.f;
        FUSG_UNROLL
        for (int a = 0; a < 6; ++a)
            FUSG_UNROLL
            for (int c = 0; c < 6; ++c) {
                float s = 0.f;
                FUSG_UNROLL
                for (int i = 0; i < 2 * nj; ++i) s += J[i][a] * J[i][c];
                A[a * 6 + c] = s;
                sum += s;
                if (a == c) dmax = fmaxf(dmax, s);
            }
        nerr = nerr < 2 ? nerr + 1 : 2;
        if (sum < 1e-7f) break;                                       // cpc.py:105-106
        if (!have_lam) { lam = 1e-8 * (double)dmax; have_lam = true; }
        float JtJ[36];
        FUSG_UNROLL
        for (int a = 0; a < 36; ++a) JtJ[a] = A[a];
        FUSG_UNROLL
        for (int a = 0; a < 6; ++a) A[a * 7] = JtJ[a * 7] + (float)lam;
        float Ai[36];
        if (!inv6(A, Ai)) break;                                      // cpc.py:116-117
        FUSG_UNROLL
        for (int a = 0; a < 6; ++a) {                                 // (-inv @ J^T) @ err, in that order (cpc.py:115)
            float s = 0.f;
            FUSG_UNROLL
            for (int i = 0; i < 2 * nj; ++i) {
                float m = 0.f;
                FUSG_UNROLL
                for (int c = 0; c < 6; ++c) m += -Ai[a * 6 + c] * J[i][c];
                s += m * err[i];
            }
            upd[a] = s;
        }
        FUSG_UNROLL
        for (int a = 0; a < 6; ++a) prm[a] += upd[a];
        it += 1;
        if (nerr == 2) {                                              // check_lambda (pnp_utils.py:27-41)
            float pcst = 0.f, ccst = 0.f, den = 0.f;
            FUSG_UNROLL
            for (int i = 0; i < 2 * n; ++i) { pcst += prev[i] * prev[i]; ccst += err[i] * err[i]; }
            pcst *= 0.5f; ccst *= 0.5f;
            FUSG_UNROLL
            for (int a = 0; a < 6; ++a) {
                float g = 0.f;
                FUSG_UNROLL
                for (int i = 0; i < 2 * nj; ++i) g += J[i][a] * err[i];
                den += upd[a] * ((float)lam * upd[a] - g);
            }
            den *= 0.5f;
            const float rho = (pcst - ccst) / den;
            if (rho <= 0.f) { lam *= factor; factor *= 2.0; }
            else {
                const double t = 2.0 * (double)rho - 1.0;
                const double m = 1.0 - t * t * t;
                lam *= m > 1.0 / 3.0 ? m : 1.0 / 3.0;                  // (a NaN gain ratio gives NaN here, as np.max does)
                if (m != m) lam = m;
                factor = 2.0;
            }
        }
    }
    float e2 = 0.f;
    FUSG_UNROLL
    for (int i = 0; i < 2 * n; ++i) e2 += err[i] * err[i];
    FUSG_UNROLL
    for (int a = 0; a < 3; ++a) { k.rvec[(long)idx * 3 + a] = prm[a]; k.tvec[(long)idx * 3 + a] = prm[3 + a]; }
    k.err[idx] = nerr ? e2 / (float)(2 * n) : __builtin_nanf("");
}

#undef FUSG_UNROLL

}  // namespace fusg

using namespace fusg;

static int pnp_impl(const float* p3, const float* p2, const float* focals, const float* centers, const float* rvec0,
                    const float* tvec0, int32_t B, int32_t n, int32_t S, int32_t max_iter, float* rvec, float* tvec,
                    float* err, void* stream) {
    FUSG_CHECK(p3 && p2 && focals && centers && rvec0 && tvec0 && rvec && tvec && err, "pnp_cpc: null pointer");
    FUSG_CHECK(B >= 0 && n >= 1 && n <= PNP_MAXP && S >= 1 && S <= 64 && max_iter >= 0 && (long)B * S < (1L << 30),
               "pnp_cpc: B %d, points %d (1..%d), starts %d (1..64), max_iter %d", B, n, PNP_MAXP, S, max_iter);
    if (B == 0) return FUSG_OK;
    PnpK k{p3, p2, focals, centers, rvec0, tvec0, rvec, tvec, err, B, n, S, max_iter};
    const int total = B * S;
    if (n == 12) hipLaunchKernelGGL(pnp_cpc_kernel<12>, dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream, k);   // the 12 keypoints
    else hipLaunchKernelGGL(pnp_cpc_kernel<0>, dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream, k);
    FUSG_LAUNCH_CHECK("pnp_cpc");
    return FUSG_OK;
}
extern "C" int fusg_pnp_cpc(const float* p3, const float* p2, const float* focals, const float* centers, const float* rvec0,
                            const float* tvec0, int32_t B, int32_t n, int32_t S, int32_t max_iter, float* rvec, float* tvec,
                            float* err, void* stream) {
    return fusg::plan_dispatch(pnp_impl, stream, p3, p2, focals, centers, rvec0, tvec0, B, n, S, max_iter, rvec, tvec, err);
}
