// libfusg: error reporting, version, and the opt-in per-kernel HIP-event profiler.
#include <stdarg.h>
#include <stdlib.h>
#include <mutex>
#include <set>
#include <utility>
#include <vector>
#include "common.h"

namespace fusg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

hipError_t ensure_dyn_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_pair(dev, fn);
    if (done.count(key)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert(key);
    return e;
}

const EnvSwitches& env_switches() {
    static const EnvSwitches sw = [] {
        EnvSwitches s;
        s.no_vec_epi = getenv("FUSG_NO_VEC_EPI") != nullptr;
        s.no_halo = getenv("FUSG_NO_HALO") != nullptr;
        s.no_touch = getenv("FUSG_NO_TOUCH") != nullptr;
        s.no_ksplit = getenv("FUSG_NO_KSPLIT") != nullptr;        // narrow halo tiles with the M split over the waves (rounds 1-2)
        s.no_pointwise = getenv("FUSG_NO_POINTWISE") != nullptr;  // 1x1 from <= 8 channels on the tap-unit MFMA kernel again          // no L2 warm-up of the weights (conv_kernel.h, l2_touch)
        s.small_maxhw = getenv("FUSG_SMALL_MAXHW") ? atoi(getenv("FUSG_SMALL_MAXHW")) : 64;   // largest Ho * Wo it takes
        s.halo_minwg = getenv("FUSG_HALO_MINWG") ? atol(getenv("FUSG_HALO_MINWG")) : 512;
        s.halo_bn = getenv("FUSG_HALO_BN") ? atoi(getenv("FUSG_HALO_BN")) : 0;
        return s;
    }();
    return sw;
}

// ---- profiler: event pairs recorded on the launch stream, resolved lazily in fusg_prof_read ----
struct ProfKind {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<hipEvent_t> pool;
    double ms = 0.0, flops = 0.0;
    long launches = 0;
    std::vector<std::pair<hipStream_t, hipEvent_t>> open;    // begin events not yet closed, by launch stream (two host
                                                             // threads or two streams may be between begin and end at once)
};
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static ProfKind g_kinds[2];

static hipEvent_t get_event(ProfKind& k) {
    if (!k.pool.empty()) { hipEvent_t e = k.pool.back(); k.pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void prof_begin(int kind, hipStream_t s, double flops) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfKind& k = g_kinds[kind];
    hipEvent_t b = get_event(k);
    k.flops += flops;
    k.launches += 1;
    (void)hipEventRecord(b, s);
    k.open.emplace_back(s, b);
}

void prof_end(int kind, hipStream_t s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfKind& k = g_kinds[kind];
    for (size_t i = k.open.size(); i-- > 0;) {
        if (k.open[i].first != s) continue;
        hipEvent_t e = get_event(k);
        (void)hipEventRecord(e, s);
        k.pending.emplace_back(k.open[i].second, e);
        k.open.erase(k.open.begin() + (long)i);
        return;
    }
}

}  // namespace fusg

using namespace fusg;

extern "C" int fusg_version(void) { return FUSG_VERSION; }
extern "C" const char* fusg_last_error(void) { return g_err; }

static thread_local int g_conv_kernel = -1;
namespace fusg { void note_conv_kernel(int kind) { g_conv_kernel = kind; } }
extern "C" int fusg_last_conv_kernel(void) { return g_conv_kernel; }
extern "C" const char* fusg_arch(void) { return "gfx950"; }
extern "C" int fusg_sizeof_tensor(void) { return (int)sizeof(fusg_tensor); }
extern "C" int fusg_sizeof_conv_desc(void) { return (int)sizeof(fusg_conv_desc); }
extern "C" int fusg_sizeof_bneck_desc(void) { return (int)sizeof(fusg_bneck_desc); }

extern "C" void fusg_prof_enable(int on) { g_prof_on = on != 0; }

extern "C" void fusg_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (ProfKind& k : g_kinds) {
        for (auto& pr : k.pending) { k.pool.push_back(pr.first); k.pool.push_back(pr.second); }
        k.pending.clear();
        for (auto& o : k.open) k.pool.push_back(o.second);
        k.open.clear();
        k.ms = 0.0; k.flops = 0.0; k.launches = 0;
    }
}

extern "C" int fusg_prof_read(int kind, double* total_ms, int64_t* launches, double* flops) {
    if (kind < 0 || kind > 1) { set_error("prof_read: kind %d", kind); return FUSG_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfKind& k = g_kinds[kind];
    for (auto& pr : k.pending) {
        if (hipEventSynchronize(pr.second) != hipSuccess) { set_error("prof_read: event sync failed"); return FUSG_ERR_LAUNCH; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) k.ms += ms;
        k.pool.push_back(pr.first);
        k.pool.push_back(pr.second);
    }
    k.pending.clear();
    if (total_ms) *total_ms = k.ms;
    if (launches) *launches = k.launches;
    if (flops) *flops = k.flops;
    return FUSG_OK;
}
