// Shared host-side helpers for libfusg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <memory>
#include <tuple>
#include <utility>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/fusg.h"

namespace fusg {

void set_error(const char* fmt, ...);
void note_conv_kernel(int kind);        // fusg_last_conv_kernel(): which kernel family the last conv launch used

#define FUSG_CHECK(cond, ...)                         \
    do {                                              \
        if (!(cond)) {                                \
            fusg::set_error(__VA_ARGS__);             \
            return FUSG_ERR_INVALID;                  \
        }                                             \
    } while (0)

#define FUSG_LAUNCH_CHECK(what)                                                    \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            fusg::set_error("%s: %s", what, hipGetErrorString(e__));               \
            return FUSG_ERR_LAUNCH;                                                \
        }                                                                          \
    } while (0)

// NHWC-physical: channels contiguous, dense pixels with channel stride Cs (= sw).
static inline bool is_nhwc(const fusg_tensor& t) {
    if (!t.data || t.dtype != FUSG_F32) return false;
    if (t.c > 1 && t.sc != 1) return false;
    if (t.sw < t.c || (t.sw & 3)) return false;
    if (t.h > 1 && t.sh != t.w * t.sw) return false;
    if (t.n > 1 && t.sn != t.h * t.w * t.sw) return false;
    if (((uintptr_t)t.data) & 15) return false;
    return true;
}

static inline bool same_nhw(const fusg_tensor& a, const fusg_tensor& b) {
    return a.n == b.n && a.h == b.h && a.w == b.w;
}
static inline bool same_shape(const fusg_tensor& a, const fusg_tensor& b) {
    return same_nhw(a, b) && a.c == b.c;
}

// One-time, per-device opt-in of a kernel to `bytes` of dynamic LDS (hipFuncSetAttribute is a driver call: it is
// made once per (device, kernel), not per launch).  Thread-safe.  (api.hip)
hipError_t ensure_dyn_lds(const void* fn, int bytes);

// Development switches, read from the environment ONCE (first use): FUSG_NO_VEC_EPI, FUSG_NO_HALO,
// FUSG_HALO_MINWG, FUSG_HALO_BN, FUSG_NO_TOUCH (tools/README.md).  (api.hip)
struct EnvSwitches { bool no_vec_epi, no_halo, no_touch, no_pointwise, no_ksplit; long halo_minwg; int halo_bn, small_maxhw; };
const EnvSwitches& env_switches();

// ---- recorded passes (plan.hip) ----------------------------------------------------------------------------------
// While a fusg_plan is recording on this thread, every launching entry point appends a closure of itself - its
// descriptors copied by value, its stream - to the plan before executing; fusg_plan_run replays the closures in order.
void plan_append(hipStream_t stream, std::function<int(hipStream_t)> fn);
bool plan_recording();

template <class A> struct SavedArg {                       // scalars and raw device pointers: by value
    A a;
    explicit SavedArg(A v) : a(v) {}
    A get() const { return a; }
};
template <> struct SavedArg<const fusg_tensor*> {          // descriptors: copied (NULL stays NULL)
    bool has; fusg_tensor t;
    explicit SavedArg(const fusg_tensor* p) : has(p != nullptr), t(p ? *p : fusg_tensor{}) {}
    const fusg_tensor* get() const { return has ? &t : nullptr; }
};
template <> struct SavedArg<const fusg_conv_desc*> {
    bool has; fusg_conv_desc d;
    explicit SavedArg(const fusg_conv_desc* p) : has(p != nullptr), d(p ? *p : fusg_conv_desc{}) {}
    const fusg_conv_desc* get() const { return has ? &d : nullptr; }
};

template <> struct SavedArg<const fusg_bneck_desc*> {
    bool has; fusg_bneck_desc d;
    explicit SavedArg(const fusg_bneck_desc* p) : has(p != nullptr), d(p ? *p : fusg_bneck_desc{}) {}
    const fusg_bneck_desc* get() const { return has ? &d : nullptr; }
};

// run impl(args..., stream); when a plan is recording, remember the call first
template <class... A>
int plan_dispatch(int (*impl)(A..., void*), void* stream, A... a) {
    if (plan_recording()) {
        auto saved = std::make_shared<std::tuple<SavedArg<A>...>>(SavedArg<A>(a)...);
        plan_append((hipStream_t)stream, [impl, saved](hipStream_t s) {
            return std::apply([&](const SavedArg<A>&... sv) { return impl(sv.get()..., (void*)s); }, *saved);
        });
    }
    return impl(a..., stream);
}

// 256 zero bytes per device, allocated on first use: where the gather of a zero-padded pixel reads (conv_igemm.hip)
const float* zero_line();

// profiler (api.hip)
void prof_begin(int kind, hipStream_t s, double flops);
void prof_end(int kind, hipStream_t s);

}  // namespace fusg
