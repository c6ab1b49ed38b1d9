// Recorded passes: one call replays a whole crop pass.
//
// The reference calls its networks at batch 1 (trajectory_inference.py:55,65), where a pass of this library is ~370
// launches of a few microseconds each and the Python / ctypes issue path (4.4 ms per pass) - not the GPU - sets the
// latency.  A fusg_plan records the launch sequence of one pass ONCE: while recording is on for the calling thread,
// every launching entry point appends a closure of itself (descriptors copied by value, stream remembered) before it
// executes (common.h: plan_dispatch), and the host adds the few things that are not libfusg launches: cross-stream
// dependencies (record an event on one stream, wait for it on another) and host-to-device copies of per-pass data
// (the VUnet's CPU-drawn noise, from a ring of pinned buffers).  fusg_plan_run then re-issues the sequence in order on the recorded streams.
//
// This is deliberately not a hipGraph: on ROCm 7.2 a captured multi-stream pass replayed 40-60 % SLOWER than eager
// launching when it was measured in round 1 (cause never established with a trace: DESIGN.md §9).  A plan keeps eager
// semantics - same streams, same priorities, same overlap - and only removes the interpreter from the issue path.
//
// Contract: every device pointer a recorded call used must stay valid and keep its meaning until the plan is destroyed
// (the Python side records inside a private torch memory pool and keeps the pool), inputs are refreshed in place.
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "common.h"

struct fusg_plan {
    struct Op {
        int kind;                                   // 0 launch closure, 1 record event, 2 wait event, 3 h2d copy
        hipStream_t stream;
        std::function<int(hipStream_t)> fn;
        hipEvent_t ev;
        void* dst; const char* src; size_t bytes, slot_stride;
        int ev0;                                    // h2d: index of this op's first per-slot event in `events`
    };
    std::vector<Op> ops;
    std::vector<hipEvent_t> events;                 // owned
    std::vector<int> h2d_ops;                       // indices of the h2d ops
    int nslots = 0;                                 // ring depth of the pinned h2d sources (same for all h2d ops)
    long runs = 0;                                  // passes issued so far (the recording counts as one)
    hipGraph_t graph = nullptr;                     // fusg_plan_graph_capture: the recording as ONE hipGraph (experiment)
    hipGraphExec_t gexec = nullptr;
    hipEvent_t gslot_ev = nullptr;                  // recorded behind every graph launch: the graph's pinned slot is free once it is reached
    int gslot = 0;
    // fusg_plan_run_mt: one issuing host thread per recorded stream
    int device = -1;                                // HIP device of the recording thread (the issuing threads make it current)
    std::vector<std::vector<int>> lanes;            // op indices per stream, in recorded order (lane 0 = the stream of the first op)
    std::vector<int> wait_src;                      // per op: for a wait, the index of the record op of its event (-1 otherwise)
    std::unique_ptr<std::atomic<long>[]> issued;    // per op: for a record, the last run whose record has been issued
    std::atomic<int> abort{0};                      // a lane failed: waiting lanes stop waiting for its records
};

namespace fusg {

static thread_local fusg_plan* g_rec = nullptr;

bool plan_recording() { return g_rec != nullptr; }

void plan_append(hipStream_t stream, std::function<int(hipStream_t)> fn) {
    if (!g_rec) return;
    fusg_plan::Op op{};
    op.kind = 0; op.stream = stream; op.fn = std::move(fn);
    g_rec->ops.push_back(std::move(op));
}

}  // namespace fusg

using namespace fusg;

extern "C" fusg_plan* fusg_plan_create(void) { return new fusg_plan(); }

extern "C" void fusg_plan_destroy(fusg_plan* p) {
    if (!p) return;
    if (g_rec == p) g_rec = nullptr;
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    if (p->gexec) (void)hipGraphExecDestroy(p->gexec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    if (p->gslot_ev) (void)hipEventDestroy(p->gslot_ev);
    delete p;
}

extern "C" int fusg_plan_begin(fusg_plan* p) {
    FUSG_CHECK(p && !g_rec, "plan_begin: null plan or a recording is already open on this thread");
    FUSG_CHECK(p->ops.empty(), "plan_begin: the plan already holds a recording");
    g_rec = p;
    return FUSG_OK;
}

extern "C" int fusg_plan_end(fusg_plan* p) {
    FUSG_CHECK(p && g_rec == p, "plan_end: this plan is not recording on this thread");
    g_rec = nullptr;
    p->runs = 1;                                    // the recording pass used slot 0
    (void)hipGetDevice(&p->device);
    return FUSG_OK;
}

// `waiter` must not run past this point before the work issued so far on `signaller` has finished.  Also performs the
// synchronisation now (the recording pass is a real pass).
extern "C" int fusg_plan_add_dependency(fusg_plan* p, void* waiter, void* signaller) {
    FUSG_CHECK(p && g_rec == p, "plan_add_dependency: plan is not recording");
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("plan_add_dependency: hipEventCreate failed"); return FUSG_ERR_LAUNCH; }
    p->events.push_back(ev);
    fusg_plan::Op rec{}; rec.kind = 1; rec.stream = (hipStream_t)signaller; rec.ev = ev;
    fusg_plan::Op wai{}; wai.kind = 2; wai.stream = (hipStream_t)waiter; wai.ev = ev;
    p->ops.push_back(rec);
    p->ops.push_back(wai);
    if (hipEventRecord(ev, (hipStream_t)signaller) != hipSuccess || hipStreamWaitEvent((hipStream_t)waiter, ev, 0) != hipSuccess) {
        set_error("plan_add_dependency: event record / wait failed");
        return FUSG_ERR_LAUNCH;
    }
    return FUSG_OK;
}

// Per-pass host data (pinned) -> device, recorded as an asynchronous copy on `stream`.  The source is a ring of
// `nslots` buffers `slot_stride` bytes apart: run r copies from slot r % nslots, so the host can fill the next slot
// while earlier passes are still queued (fusg_plan_next_slot tells which, and waits until that slot's last copy has
// executed).  Also copies now, from slot 0 (the recording is run 0).
extern "C" int fusg_plan_add_h2d(fusg_plan* p, void* dst, const void* src, int64_t bytes, int32_t nslots, int64_t slot_stride, void* stream) {
    FUSG_CHECK(p && g_rec == p && dst && src && bytes > 0 && nslots >= 1 && slot_stride >= bytes, "plan_add_h2d: plan is not recording / bad arguments");
    FUSG_CHECK(p->nslots == 0 || p->nslots == nslots, "plan_add_h2d: all copies of a plan use the same ring depth (%d)", p->nslots);
    p->nslots = nslots;
    fusg_plan::Op op{};
    op.kind = 3; op.stream = (hipStream_t)stream; op.dst = dst; op.src = (const char*)src; op.bytes = (size_t)bytes;
    op.slot_stride = (size_t)slot_stride; op.ev0 = (int)p->events.size();
    for (int i = 0; i < nslots; ++i) {
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("plan_add_h2d: hipEventCreate failed"); return FUSG_ERR_LAUNCH; }
        p->events.push_back(ev);
    }
    p->h2d_ops.push_back((int)p->ops.size());
    p->ops.push_back(op);
    if (hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipEventRecord(p->events[op.ev0], (hipStream_t)stream) != hipSuccess) {
        set_error("plan_add_h2d: copy failed");
        return FUSG_ERR_LAUNCH;
    }
    return FUSG_OK;
}

// Slot of the pinned rings that the NEXT fusg_plan_run will copy from; returns once the copies that last used that
// slot (nslots passes ago) have executed, so that the host may overwrite it.  -1 on error.
extern "C" int fusg_plan_next_slot(fusg_plan* p) {
    if (!p || g_rec) { set_error("plan_next_slot: null plan or still recording"); return -1; }
    if (p->nslots == 0) return 0;
    const int slot = (int)(p->runs % p->nslots);
    if (p->runs >= p->nslots)
        for (int oi : p->h2d_ops)
            if (hipEventSynchronize(p->events[p->ops[oi].ev0 + slot]) != hipSuccess) { set_error("plan_next_slot: event sync failed"); return -1; }
    return slot;
}

extern "C" int64_t fusg_plan_size(const fusg_plan* p) { return p ? (int64_t)p->ops.size() : -1; }

extern "C" int fusg_plan_run(fusg_plan* p) {
    FUSG_CHECK(p && !g_rec, "plan_run: null plan or a recording is open on this thread");
    const int slot = p->nslots ? (int)(p->runs % p->nslots) : 0;
    for (fusg_plan::Op& op : p->ops) {
        switch (op.kind) {
            case 0: { const int rc = op.fn(op.stream); if (rc != FUSG_OK) return rc; break; }
            case 1: if (hipEventRecord(op.ev, op.stream) != hipSuccess) { set_error("plan_run: event record failed"); return FUSG_ERR_LAUNCH; } break;
            case 2: if (hipStreamWaitEvent(op.stream, op.ev, 0) != hipSuccess) { set_error("plan_run: stream wait failed"); return FUSG_ERR_LAUNCH; } break;
            default:
                if (hipMemcpyAsync(op.dst, op.src + (size_t)slot * op.slot_stride, op.bytes, hipMemcpyHostToDevice, op.stream) != hipSuccess ||
                    hipEventRecord(p->events[op.ev0 + slot], op.stream) != hipSuccess) { set_error("plan_run: h2d copy failed"); return FUSG_ERR_LAUNCH; }
        }
    }
    p->runs += 1;
    return FUSG_OK;
}


// ---- multi-threaded replay ------------------------------------------------------------------------------------------
// At small batches the GPU finishes a pass faster than ONE host thread can issue it: at batch 1 the 299 operations of a crop pass
// take 1.98 ms to issue (6.6 us each: hipLaunchKernel) and 2.02 ms to run (profiles/r04_small_batch.jsonl) - the pass is bound by the
// issuing thread, and so is the frame driver (5.8 ms of issue for a 7.4 ms frame).  A pass runs on 4-5 HIP streams
// (one per network branch), and the runtime's launch path is per stream: fusg_plan_run_mt issues every recorded stream from a host
// thread of its own - the caller's thread takes the first stream, persistent worker threads the others.  Order inside a stream is the
// recorded order; across streams the only ordering a recording has are its dependencies (event record on one stream, wait on another),
// and a waiting thread holds its hipStreamWaitEvent back until the recording thread has ISSUED that run's hipEventRecord (an atomic
// run counter per record operation) - a wait issued before its record would wait for the previous pass's.  Same launches, same
// streams, same results as fusg_plan_run.
namespace {

inline void cpu_relax() { __builtin_ia32_pause(); }

struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> state{0};                      // 0 idle, 1 job posted, 2 job done
    bool asleep = false;
    fusg_plan* p = nullptr;
    int lane = 0, slot = 0, rc = 0;
    long run = 0;
    char err[512] = "";
};

int run_lane(fusg_plan* p, int lane, long run, int slot) {
    for (int oi : p->lanes[lane]) {
        fusg_plan::Op& op = p->ops[oi];
        switch (op.kind) {
            case 0: { const int rc = op.fn(op.stream); if (rc != FUSG_OK) return rc; break; }
            case 1:
                if (hipEventRecord(op.ev, op.stream) != hipSuccess) { set_error("plan_run_mt: event record failed"); return FUSG_ERR_LAUNCH; }
                p->issued[oi].store(run, std::memory_order_release);
                break;
            case 2: {
                std::atomic<long>& a = p->issued[p->wait_src[oi]];
                while (a.load(std::memory_order_acquire) < run) {
                    if (p->abort.load(std::memory_order_relaxed)) { set_error("plan_run_mt: another stream's issue failed"); return FUSG_ERR_LAUNCH; }
                    cpu_relax();
                }
                if (hipStreamWaitEvent(op.stream, op.ev, 0) != hipSuccess) { set_error("plan_run_mt: stream wait failed"); return FUSG_ERR_LAUNCH; }
                break;
            }
            default:
                if (hipMemcpyAsync(op.dst, op.src + (size_t)slot * op.slot_stride, op.bytes, hipMemcpyHostToDevice, op.stream) != hipSuccess ||
                    hipEventRecord(p->events[op.ev0 + slot], op.stream) != hipSuccess) { set_error("plan_run_mt: h2d copy failed"); return FUSG_ERR_LAUNCH; }
        }
    }
    return FUSG_OK;
}

void worker_main(Worker* w) {
    for (;;) {
        // a pass follows a pass within microseconds in a frame loop: spin for a while before going to sleep
        int spins = 0;
        while (w->state.load(std::memory_order_acquire) != 1) {
            if (++spins < 20000) { cpu_relax(); continue; }
            std::unique_lock<std::mutex> lk(w->mu);
            w->asleep = true;
            w->cv.wait(lk, [w] { return w->state.load(std::memory_order_acquire) == 1; });
            w->asleep = false;
        }
        (void)hipSetDevice(w->p->device);
        w->rc = run_lane(w->p, w->lane, w->run, w->slot);
        if (w->rc != FUSG_OK) {
            snprintf(w->err, sizeof w->err, "%s", fusg_last_error());
            w->p->abort.store(1, std::memory_order_relaxed);
        }
        w->state.store(2, std::memory_order_release);
    }
}

std::mutex g_pool_mu;                               // one multi-threaded replay at a time (the workers are shared by all plans)
std::vector<Worker*> g_pool;                        // never destroyed: the threads end with the process

}  // namespace

extern "C" int32_t fusg_plan_streams(fusg_plan* p) {
    if (!p || g_rec == p || p->ops.empty()) return -1;
    if (p->lanes.empty()) {
        std::vector<hipStream_t> streams;
        p->wait_src.assign(p->ops.size(), -1);
        p->issued.reset(new std::atomic<long>[p->ops.size()]);
        for (size_t i = 0; i < p->ops.size(); ++i) {
            p->issued[i].store(0, std::memory_order_relaxed);
            const fusg_plan::Op& op = p->ops[i];
            size_t l = 0;
            while (l < streams.size() && streams[l] != op.stream) ++l;
            if (l == streams.size()) { streams.push_back(op.stream); p->lanes.emplace_back(); }
            p->lanes[l].push_back((int)i);
            if (op.kind == 2) {                     // its record: the latest earlier record op of the same event
                for (size_t j = i; j-- > 0;)
                    if (p->ops[j].kind == 1 && p->ops[j].ev == op.ev) { p->wait_src[i] = (int)j; break; }
                if (p->wait_src[i] < 0) { p->lanes.clear(); set_error("plan_streams: a wait without an earlier record"); return -1; }
            }
        }
    }
    return (int32_t)p->lanes.size();
}

extern "C" int fusg_plan_run_mt(fusg_plan* p) {
    FUSG_CHECK(p && !g_rec, "plan_run_mt: null plan or a recording is open on this thread");
    std::lock_guard<std::mutex> pool_lock(g_pool_mu);                 // (also guards the lazy stream split of fusg_plan_streams)
    const int nl = fusg_plan_streams(p);
    FUSG_CHECK(nl >= 1, "plan_run_mt: empty plan");
    if (nl == 1) return fusg_plan_run(p);
    while ((int)g_pool.size() < nl - 1) {
        Worker* w = new Worker();
        w->th = std::thread(worker_main, w);
        w->th.detach();
        g_pool.push_back(w);
    }
    const int slot = p->nslots ? (int)(p->runs % p->nslots) : 0;
    const long run = p->runs + 1;
    p->abort.store(0, std::memory_order_relaxed);
    for (int l = 1; l < nl; ++l) {
        Worker* w = g_pool[l - 1];
        w->p = p; w->lane = l; w->slot = slot; w->run = run; w->rc = FUSG_OK;
        bool wake;
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->state.store(1, std::memory_order_release);
            wake = w->asleep;
        }
        if (wake) w->cv.notify_one();
    }
    int rc = run_lane(p, 0, run, slot);
    char err[512] = "";
    if (rc != FUSG_OK) { snprintf(err, sizeof err, "%s", fusg_last_error()); p->abort.store(1, std::memory_order_relaxed); }
    for (int l = 1; l < nl; ++l) {
        Worker* w = g_pool[l - 1];
        while (w->state.load(std::memory_order_acquire) != 2) cpu_relax();
        if (w->rc != FUSG_OK && rc == FUSG_OK) { rc = w->rc; snprintf(err, sizeof err, "%s", w->err); }
        w->state.store(0, std::memory_order_release);
    }
    p->runs += 1;
    if (rc != FUSG_OK) set_error("%s", err);
    return rc;
}


// ---- the same recording as ONE hipGraph ---------------------------------------------------------------------------
// Re-issues the recorded sequence under a stream capture that starts on `capture_stream` (the stream the pass was recorded
// from: the recorded cross-stream dependencies pull the side streams into the capture and join them back) and instantiates
// the captured graph.  Everything that may not happen under capture stays out of it: the per-slot event records of the h2d
// ring (a captured event cannot be synchronised from the host) - the graph copies from ONE pinned slot, `slot`, and
// fusg_plan_graph_launch orders the host against it with an event recorded OUTSIDE the graph.  The capture is ALWAYS ended,
// also when an operation fails inside it, so a failed attempt leaves no stream in capture mode (round 3's experiment did
// not, and torch's allocator asserted at exit).  Thread-local capture mode: other threads' HIP calls are unaffected.
// Returns FUSG_OK, or an error whose text names the recorded operation (index, kind) that failed.
extern "C" int fusg_plan_graph_capture(fusg_plan* p, void* capture_stream, int32_t slot) {
    FUSG_CHECK(p && !g_rec && !p->ops.empty(), "plan_graph_capture: null / empty plan or a recording is open");
    FUSG_CHECK(!p->gexec, "plan_graph_capture: this plan already holds a graph");
    FUSG_CHECK(slot >= 0 && (p->nslots == 0 || slot < p->nslots), "plan_graph_capture: slot %d of %d", slot, p->nslots);
    hipStream_t cs = (hipStream_t)capture_stream;
    if (hipEventCreateWithFlags(&p->gslot_ev, hipEventDisableTiming) != hipSuccess) { set_error("plan_graph_capture: hipEventCreate failed"); return FUSG_ERR_LAUNCH; }
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { set_error("plan_graph_capture: hipStreamBeginCapture: %s", hipGetErrorString(e)); return FUSG_ERR_LAUNCH; }
    int rc = FUSG_OK, bad = -1;
    char why[256] = "";
    for (size_t i = 0; i < p->ops.size() && rc == FUSG_OK; ++i) {
        fusg_plan::Op& op = p->ops[i];
        hipError_t he = hipSuccess;
        switch (op.kind) {
            case 0: rc = op.fn(op.stream); if (rc != FUSG_OK) snprintf(why, sizeof why, "launch closure: %s", fusg_last_error()); break;
            case 1: he = hipEventRecord(op.ev, op.stream); break;
            case 2: he = hipStreamWaitEvent(op.stream, op.ev, 0); break;
            default: he = hipMemcpyAsync(op.dst, op.src + (size_t)slot * op.slot_stride, op.bytes, hipMemcpyHostToDevice, op.stream);
        }
        if (he != hipSuccess) { rc = FUSG_ERR_LAUNCH; snprintf(why, sizeof why, "%s", hipGetErrorString(he)); }
        if (rc != FUSG_OK) bad = (int)i;
    }
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(cs, &g);                               // always: a failed attempt must not leave the stream capturing
    (void)hipGetLastError();
    if (rc != FUSG_OK) {
        if (g) (void)hipGraphDestroy(g);
        set_error("plan_graph_capture: recorded operation %d (kind %d) failed under capture: %s", bad, p->ops[bad].kind, why);
        return rc;
    }
    if (e != hipSuccess || !g) { set_error("plan_graph_capture: hipStreamEndCapture: %s", hipGetErrorString(e)); if (g) (void)hipGraphDestroy(g); return FUSG_ERR_LAUNCH; }
    hipGraphExec_t ge = nullptr;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { set_error("plan_graph_capture: hipGraphInstantiate: %s", hipGetErrorString(e)); (void)hipGraphDestroy(g); return FUSG_ERR_LAUNCH; }
    p->graph = g; p->gexec = ge; p->gslot = slot;
    return FUSG_OK;
}

extern "C" int64_t fusg_plan_graph_nodes(const fusg_plan* p) {
    if (!p || !p->graph) return -1;
    size_t n = 0;
    if (hipGraphGetNodes(p->graph, nullptr, &n) != hipSuccess) return -1;
    return (int64_t)n;
}

// Waits until the previous graph launch has consumed the graph's pinned slot (so the host may refill it), then returns the slot.
extern "C" int fusg_plan_graph_slot(fusg_plan* p) {
    if (!p || !p->gexec) { set_error("plan_graph_slot: no graph"); return -1; }
    if (hipEventSynchronize(p->gslot_ev) != hipSuccess) { set_error("plan_graph_slot: event sync failed"); return -1; }
    return p->gslot;
}

extern "C" int fusg_plan_graph_launch(fusg_plan* p, void* stream) {
    FUSG_CHECK(p && p->gexec, "plan_graph_launch: no graph (fusg_plan_graph_capture first)");
    hipError_t e = hipGraphLaunch(p->gexec, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("plan_graph_launch: %s", hipGetErrorString(e)); return FUSG_ERR_LAUNCH; }
    if (hipEventRecord(p->gslot_ev, (hipStream_t)stream) != hipSuccess) { set_error("plan_graph_launch: event record failed"); return FUSG_ERR_LAUNCH; }
    return FUSG_OK;
}
