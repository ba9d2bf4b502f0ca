// torch_ext.cpp -- the PyTorch-ROCm side of the native step executor: autograd node, device buffers and streams for
// mccnn_geometry_* / mccnn_conv_* (include/mccnn.h), so that a convolution costs the host ONE Python -> C++ call in the
// forward pass and NONE in the backward pass (the autograd engine calls the C++ node directly).
//
// torch only supplies what the C-ABI leaves to its caller: device memory (at::empty -- the library never allocates), the
// current HIP stream and the autograd graph. Every kernel launch goes through libmccnn_hip.so. Built by
// mccnn_amd/build.py with g++ against the torch headers (host code only, no kernels here); mccnn_amd/native.py falls
// back to the ctypes form of the same calls when this module is not built.
#include <torch/extension.h>
#include <execinfo.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/functions/utils.h>
#include <torch/csrc/autograd/saved_variable.h>
#include <c10/hip/HIPStream.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <map>
#include <thread>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "mccnn.h"
#include "debug_opts.h"

namespace {

using at::Tensor;

struct CapacityError : std::runtime_error {
    int edges;
    CapacityError(int e) : std::runtime_error("neighbour list longer than the geometry's capacity"), edges(e) {}
};

void check(int rc, const char* what) {
    if (rc == 0) return;
    throw std::runtime_error(std::string(what) + " failed: " + mccnn_error_string(rc) + " (code " + std::to_string(rc) + ")");
}

// The HIP runtime's current device for the length of a call: the library launches on streams of its tensors' device and
// that device has to be current on the launching thread (a caller working with several devices may have another one set).
struct DevGuard {
    int prev = 0;
    bool changed = false;
    explicit DevGuard(int d) {
        if (hipGetDevice(&prev) == hipSuccess && d != prev && d >= 0) changed = hipSetDevice(d) == hipSuccess;
    }
    ~DevGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};

void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
// a raw stream as PyTorch's allocator knows it (pool selection: HIPStreamGuard; consumers: Tensor::record_stream)
c10::Stream as_torch_stream(void* s, int dev) {
    return c10::hip::getStreamFromExternalMasqueradingAsCUDA((hipStream_t)s, (c10::DeviceIndex)dev).unwrap();
}

hipEvent_t take_event_fwd();
void give_event_fwd(hipEvent_t e);

// grow-only scratch per (thread, device, stream): consecutive calls on a stream are stream-ordered and may share it
Tensor& scratch(size_t bytes, const Tensor& like, void* stream) {
    thread_local std::map<std::pair<int, void*>, Tensor> pool;
    Tensor& t = pool[{(int)like.device().index(), stream}];
    if (bytes < 256) bytes = 256;
    if (!t.defined() || (size_t)t.numel() < bytes) {
        // the old block goes back to the allocator of the CALLING thread's current stream, which need not be `stream`:
        // what `stream` still runs on it has to be over first (rare: the scratch only grows)
        if (t.defined()) (void)hipStreamSynchronize((hipStream_t)stream);
        t = at::empty({(int64_t)(bytes + bytes / 4)}, like.options().dtype(at::kByte));
        // ... and the NEW block comes from the caching allocator's pool of the calling thread's current stream: the
        // allocator hands out a block its last owner freed a moment ago, because work enqueued on THAT stream is ordered
        // behind the old owner's. Work on `stream` is not -- a helper thread's scratch used to be written on its side
        // stream while convolution kernels still read the block as THEIR scratch (wrong level sizes of a prefetched
        // hierarchy, wrong outputs: tools/soak_network.py SOAK_CFG=cfg4 SOAK_DEEP=1). `stream` first waits for what
        // the allocation stream holds now.
        void* alloc = cur_stream(like);
        if (alloc != stream) {
            hipEvent_t ev = take_event_fwd();
            if (hipEventRecord(ev, (hipStream_t)alloc) == hipSuccess) (void)hipStreamWaitEvent((hipStream_t)stream, ev, 0);
            give_event_fwd(ev);
        }
    }
    return t;
}

// pinned words the count pass stores the edge total into (device-accessible host memory: no copy is enqueued)
std::mutex g_slot_mutex;
std::vector<Tensor> g_slots;
void poll_parked();
Tensor take_slot() {
    poll_parked();
    {
        std::lock_guard<std::mutex> lk(g_slot_mutex);
        if (!g_slots.empty()) {
            Tensor t = g_slots.back();
            g_slots.pop_back();
            return t;
        }
    }
    return at::empty({1}, at::TensorOptions().dtype(at::kInt).pinned_memory(true));
}
void give_slot(Tensor t) {
    std::lock_guard<std::mutex> lk(g_slot_mutex);
    if (g_slots.size() < 256) g_slots.push_back(std::move(t));
}
// A geometry destroyed before its edge total arrived (dropped right after being queued, or on an exception path): the count
// pass may still WRITE the word, so the slot is neither reused nor returned to the host allocator until the build has
// retired -- parked with an event recorded behind the build, polled whenever a slot is taken.
hipEvent_t take_event();
void give_event(hipEvent_t e);
struct ParkedSlot { Tensor slot; hipEvent_t ev; };
std::vector<ParkedSlot> g_parked;   // (under g_slot_mutex)
void park_slot(Tensor t, hipEvent_t ev) {
    std::lock_guard<std::mutex> lk(g_slot_mutex);
    g_parked.push_back({std::move(t), ev});
}
void poll_parked() {
    std::vector<hipEvent_t> done;
    {
        std::lock_guard<std::mutex> lk(g_slot_mutex);
        for (size_t k = 0; k < g_parked.size();) {
            ParkedSlot& p = g_parked[k];
            const bool arrived = *reinterpret_cast<volatile int*>(p.slot.data_ptr()) >= 0;
            if (arrived || !p.ev || hipEventQuery(p.ev) == hipSuccess) {
                if (p.ev) done.push_back(p.ev);
                if (g_slots.size() < 256) g_slots.push_back(std::move(p.slot));
                g_parked[k] = std::move(g_parked.back());
                g_parked.pop_back();
            } else {
                ++k;
            }
        }
    }
    for (hipEvent_t e : done) give_event(e);
}

// Side streams for geometry builds issued ahead of their first use (ConvolutionBuilder's learned prefetch): the chains of
// a step's geometries are independent of each other and of the features, so they run side by side -- each is a dozen
// small dependent kernels that leave the chip nearly empty -- and the layer that consumes one waits for its event.
constexpr int kSideStreams = 4;
constexpr int kMaxDevices = 16;
int device_now() {
    int d = 0;
    (void)hipGetDevice(&d);
    return (d >= 0 && d < kMaxDevices) ? d : 0;
}
// (streams of the CURRENT device: every entry point below runs under a guard of its tensors' device)
hipStream_t side_stream(int k) {
    static hipStream_t streams[kMaxDevices][kSideStreams] = {};
    static bool made[kMaxDevices] = {};
    static std::mutex m;
    const int d = device_now();
    std::lock_guard<std::mutex> lk(m);
    if (!made[d]) {
        for (int i = 0; i < kSideStreams; ++i)
            if (hipStreamCreateWithFlags(&streams[d][i], hipStreamNonBlocking) != hipSuccess) streams[d][i] = nullptr;
        made[d] = true;
    }
    return streams[d][((k % kSideStreams) + kSideStreams) % kSideStreams];
}
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// events are pooled: creating and destroying one per geometry cost more than the record and the wait together
std::mutex g_event_mutex;
std::vector<hipEvent_t> g_events;
hipEvent_t take_event() {
    {
        std::lock_guard<std::mutex> lk(g_event_mutex);
        if (!g_events.empty()) {
            hipEvent_t e = g_events.back();
            g_events.pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    return e;
}
void give_event(hipEvent_t e) {
    std::lock_guard<std::mutex> lk(g_event_mutex);
    if (g_events.size() < 256) g_events.push_back(e);
    else (void)hipEventDestroy(e);
}
hipEvent_t take_event_fwd() { return take_event(); }
void give_event_fwd(hipEvent_t e) { give_event(e); }

// A helper thread starts with device 0 current; every job names the device of its tensors first (one process per GPU sets
// its device on the calling thread only -- kernel launches, memsets and event records of a job go to streams of THAT
// device and need it current on the issuing thread as well).
void enter_device(int dev) {
    thread_local int current = -1;
    if (dev != current && hipSetDevice(dev) == hipSuccess) current = dev;
}

// The launches of a side-stream build are ISSUED by a helper thread: a step's geometries are a dozen launches each
// (~35 us of host time), and the calling thread has the layers to issue. One thread, jobs in order (a geometry that
// shares another one's grid is queued behind it). MCCNN_ISSUE_THREAD=0: the calling thread issues them itself.
thread_local bool t_helper_thread = false;

class Issuer {
public:
    // 0: geometry builds of the step in flight; 1: point hierarchies of the NEXT batch (those jobs block on read-backs);
    // 2: row plans / transposed lists of the step's geometries (those jobs wait for edge totals)
    static Issuer& get(int which = 0) {
        static Issuer inst[3];
        inst[which].which_ = which;
        return inst[which];
    }
    static bool enabled() {
        static const bool on = mccnn::debug_int("issue_thread", 1) != 0;
        return on;
    }
    // (debugging: MCCNN_ISSUE_INLINE = mask of the issuers whose jobs run on the calling thread instead)
    static bool inline_jobs(int which) {
        static const int mask = mccnn::debug_int("issue_inline", 0);
        return (mask >> which) & 1;
    }
    int which_ = 0;
    void push(std::function<void()> job) {
        bool queued = false;
        if (!inline_jobs(which_)) {
            std::lock_guard<std::mutex> lk(m_);
            if (!stop_) {   // (retired: the job runs on the caller's thread)
                if (!started_) {
                    th_ = std::thread([this] { run(); });
                    started_ = true;
                }
                q_.push_back(std::move(job));
                queued = true;
            }
        }
        if (queued) cv_.notify_one();
        else job();
    }
    // Runs the queued jobs to the end and joins the thread; later jobs run on the caller's thread. Called from Python's
    // atexit (shutdown_helpers): a job -- or its destruction -- may drop the last reference to a tensor that has a Python
    // object, which takes the GIL; on a helper thread during interpreter finalisation that attempt ends the thread with a
    // forced unwind (std::terminate). After this no helper thread exists that could hold such a reference.
    void retire() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        if (started_ && th_.joinable()) th_.join();
    }
    ~Issuer() { retire(); }

private:
    void run() {
        t_helper_thread = true;
        mccnn_debug_wait_accounting(0);
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                job = std::move(q_.front());
                q_.pop_front();
            }
            static const int delay_us = mccnn::debug_int("job_delay_us", 0);
            if (delay_us > 0) {   // fault injection of the soak tests: a random pause of 0 .. delay_us before every job
                static thread_local unsigned lcg = 12345u + (unsigned)(uintptr_t)this;
                lcg = lcg * 1664525u + 1013904223u;
                std::this_thread::sleep_for(std::chrono::microseconds((lcg >> 8) % (unsigned)(delay_us + 1)));
            }
            job();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    std::thread th_;
    bool started_ = false, stop_ = false;
};

void count_wait(std::chrono::steady_clock::time_point t0);

// how often a geometry nobody joined ordered its CALLER's stream behind its kernels when it died (debug_counters(): the
// regression check of NOTES round-6 item 6c -- zero would mean the waits are being skipped again)
std::atomic<long long> g_caller_orderings{0};

struct Geo {
    mccnn_geometry_t* h = nullptr;
    std::atomic<int> issued{1};    // 0 while the helper thread has not issued this geometry's build yet
    int build_rc = 0;
    hipEvent_t event = nullptr;    // recorded behind a build on a side stream; its consumers wait for it once
    bool needs_wait = false;
    int side = -1;                 // index of the side stream the build ran on (-1: the caller's stream)
    void* alloc_stream = nullptr;  // the stream the buffers were allocated on (the caller's at build time)
    // own_pool: `buf` and the prebuilt pieces come from the caching allocator's pool of the build's OWN side stream (a
    // block it hands out was last used on that stream: the build needs no ordering behind the calling stream for its
    // memory) and the build waits for the event of the prefetched hierarchy it reads instead of for everything the
    // calling stream holds -- the next batch's geometry then starts when it is asked for, not when the device has
    // finished the step in flight. Every stream that joins the build is made known to the allocator (record_stream).
    bool own_pool = false;
    void* caller_stream = nullptr;  // own_pool: the stream of the thread that asked for the build (where its inputs are consumed)
    // (... which is a NULL handle when that thread runs on the device's default stream -- the usual case: `has_caller`, not the
    // handle, says whether it is set. Round 6: the waits below were skipped for exactly that stream, and a loop that prefetched
    // geometry it never consumed let the next hierarchy overwrite level tensors that builds / plan kernels were still reading.)
    bool has_caller = false;
    void order_caller_behind(hipEvent_t ev) {
        if (!(own_pool && has_caller && caller_waits())) return;
        g_caller_orderings.fetch_add(1, std::memory_order_relaxed);
        (void)hipStreamWaitEvent((hipStream_t)caller_stream, ev, 0);
    }
    static bool caller_waits() { static const bool on = mccnn::debug_int("caller_join_off", 0) == 0; return on; }   // (fault injection: the round-6 bug back)
    Tensor buf, slot;
    std::vector<Tensor> keep, attached;
    std::shared_ptr<Geo> grid_owner;
    int n = 0, m = 0, nc = 0, B = 0;
    int64_t e_cap = 0;
    std::atomic<int> e{-1};   // (also set by the helper thread that reads the total for the pieces it issues)
    int uses = 0;  // layers convolved over this geometry so far (the builder counts)
    std::atomic<int> pieces_issued{1};  // 0 while the helper thread still has pieces of this geometry to attach / issue
    void wait_build_issued_nothrow() {
        int spins = 0;
        while (!issued.load(std::memory_order_acquire))
            if (++spins > 2000) std::this_thread::yield();
    }
    void wait_issued_nothrow() {
        if (issued.load(std::memory_order_acquire) && pieces_issued.load(std::memory_order_acquire)) return;
        // (time spent here is a WAIT for a helper thread, which in turn waits for the device: counted with the waits for
        // device-side sizes, not with the calling thread's own work -- wait_ns())
        const auto t0 = std::chrono::steady_clock::now();
        wait_build_issued_nothrow();
        int spins = 0;
        while (!pieces_issued.load(std::memory_order_acquire))
            if (++spins > 2000) std::this_thread::yield();
        count_wait(t0);
    }
    // the build's launches (and its event record) have been issued, and so have the pieces asked for with it: nothing
    // of the handle is touched before
    void wait_issued() {
        wait_issued_nothrow();
        if (build_rc) {
            const int rc = build_rc;
            build_rc = 0;
            check(rc, "geometry_build");
        }
    }
    // order `stream` behind the side-stream build (once: every later use of the geometry is on that stream as well)
    // Pieces prebuilt on the side stream come in two stages with an event each: the forward row plan (and the records),
    // then the transposed list and the transposed row plan. A forward pass waits for the first only -- unless it would
    // have to build something itself (no forward plan among the pieces, another `avg`), which must not overlap the
    // second stage: both write the shared records. A backward pass waits for both.
    void join(void* stream, bool everything = true) {
        wait_issued();
        bool joined = false;
        if (needs_wait && event) {
            hip_check(hipStreamWaitEvent((hipStream_t)stream, event, 0), "hipStreamWaitEvent");
            needs_wait = false;
            joined = true;
        }
        if (plan_wait && plan_event) {
            hip_check(hipStreamWaitEvent((hipStream_t)stream, plan_event, 0), "hipStreamWaitEvent");
            plan_wait = false;
            joined = true;
        }
        if (tr_wait && tr_event && (everything || !(have & 1))) {
            hip_check(hipStreamWaitEvent((hipStream_t)stream, tr_event, 0), "hipStreamWaitEvent");
            tr_wait = false;
            joined = true;
        }
        if (joined && own_pool && stream != alloc_stream && buf.defined()) {
            const c10::Stream consumer = as_torch_stream(stream, (int)buf.device().index());
            buf.record_stream(consumer);
            for (Tensor& t : attached) t.record_stream(consumer);
            // (the inputs the library borrowed pointers of: a consumer on a stream other than the one that made the geometry
            // reads through them too -- the grid's sorted copies are ours, the centres are the caller's)
            if (stream != caller_stream)
                for (Tensor& t : keep) if (t.defined()) t.record_stream(consumer);
        }
    }
    ~Geo() {
        wait_issued_nothrow();
        const bool total_pending = slot.defined() && e.load(std::memory_order_relaxed) < 0 &&
                                   *reinterpret_cast<volatile int*>(slot.data_ptr()) < 0;
        hipEvent_t slot_ev = nullptr;
        // a grid shared from a geometry on ANOTHER side stream's pool: its memory may go back to that pool right after this
        // (the owner outlives its users), and this build's reads of it are ordered on this stream only
        if (event && build_rc == 0 && side >= 0 && grid_owner && grid_owner->own_pool && grid_owner->side != side)
            (void)hipStreamWaitEvent((hipStream_t)grid_owner->alloc_stream, event, 0);
        if (event) {
            // never consumed: the buffers go back to the allocator of the stream they were taken on -- order that
            // stream behind the build first, or the next owner of the memory could race with it
            // (this destructor may run on the helper thread: the allocation stream is the one remembered at build time)
            if (needs_wait && buf.defined() && build_rc == 0) {
                (void)hipStreamWaitEvent((hipStream_t)alloc_stream, event, 0);
                // own pool: the INPUTS (`keep`: tensors of the hierarchy, freed after this) are known to the allocator as
                // used on the caller's stream only -- a build nobody joined has to be over before that stream moves on
                order_caller_behind(event);
            }
            if (total_pending && build_rc == 0) slot_ev = event;   // (the event of the build: the parked slot keeps it)
            else give_event(event);
        } else if (total_pending && build_rc == 0 && buf.defined()) {
            // built on the caller's stream: an event recorded there now lies behind the count pass
            try {
                slot_ev = take_event();
                if (hipEventRecord(slot_ev, (hipStream_t)alloc_stream) != hipSuccess) { give_event(slot_ev); slot_ev = nullptr; }
            } catch (const std::exception&) {
                slot_ev = nullptr;
            }
        }
        if (plan_event) {
            if (plan_wait && buf.defined()) {
                (void)hipStreamWaitEvent((hipStream_t)alloc_stream, plan_event, 0);
                order_caller_behind(plan_event);
            }
            give_event(plan_event);
        }
        if (tr_event) {
            if (tr_wait && buf.defined()) {
                (void)hipStreamWaitEvent((hipStream_t)alloc_stream, tr_event, 0);
                order_caller_behind(tr_event);
            }
            give_event(tr_event);
        }
        if (h) mccnn_geometry_destroy(h);
        if (slot.defined()) {
            if (!total_pending) give_slot(std::move(slot));
            else if (slot_ev) park_slot(std::move(slot), slot_ev);   // the count pass may still write the word
            else if (build_rc != 0) give_slot(std::move(slot));     // the build never ran: nobody writes it
            else park_slot(std::move(slot), nullptr);                // (no event to be had: recycled on the next poll)
        }
    }
    hipEvent_t plan_event = nullptr;   // recorded behind the forward row plan prebuilt on a side stream (prebuild)
    bool plan_wait = false;
    hipEvent_t tr_event = nullptr;     // ... behind the transposed list / transposed row plan
    bool tr_wait = false;
    int pre_avg = -1;                  // the `avg` the prebuilt plans were made for
    int have = 0;                      // pieces attached so far (mask)
    int plan_side = -1;                // batch builds: the side stream this geometry's row plans / transposed list go to (spread over all of them)

    // The attached pieces of `what` (1 | 2 | 4) on stream ss, forward stage first; one event per stage.
    int issue_pieces(int what, bool avg, hipStream_t ss, void* ws, size_t wsb) {
        int rc = 0;
        if (what & 1) {
            rc = mccnn_geometry_prebuild(h, 1, avg ? 1 : 0, ws, wsb, (void*)ss);
            if (!rc) {
                if (!plan_event) plan_event = take_event();
                if (hipEventRecord(plan_event, ss) != hipSuccess) rc = (int)hipErrorUnknown;
                plan_wait = true;
            }
        }
        if (!rc && (what & 6)) {
            rc = mccnn_geometry_prebuild(h, what & 6, avg ? 1 : 0, ws, wsb, (void*)ss);
            if (!rc) {
                if (!tr_event) tr_event = take_event();
                if (hipEventRecord(tr_event, ss) != hipSuccess) rc = (int)hipErrorUnknown;
                tr_wait = true;
            }
        }
        pre_avg = avg ? 1 : 0;
        return rc;
    }

    // Builds the pieces of `what` (1 forward row plan, 2 transposed row plan, 4 transposed list; the per-edge records
    // come with a plan) on side stream `side_k`, behind this geometry's own build and behind whatever the calling stream
    // holds now (`like`'s stream: the buffers are allocated there). Waits for the edge total. The layers that use the
    // geometry order their stream behind the pieces' event.
    void prebuild(int what, bool avg, int side_k, const Tensor& like) {
        const DevGuard device_guard((int)like.device().index());
        wait_issued();
        const int E = edges(-1);
        if (E <= 0 || E > e_cap) return;   // (an overflowing list is rebuilt by the first layer: nothing to build ahead)
        if (what & 3) what |= 8;           // a plan permutes the per-edge records
        if (what & 2) what |= 4;           // the transposed plan is laid out over the transposed list
        what &= ~have;
        if (!what) return;
        void* main_stream = cur_stream(like);
        hipStream_t ss = side >= 0 ? side_stream(side) : side_stream(side_k);   // behind the build: its own side stream
        if (!ss) return;
        size_t wsb = 256;
        for (int bit = 1; bit <= 8; bit <<= 1) {
            if (!(what & bit)) continue;
            long long bytes = 0, w = 0;
            check(mccnn_geometry_piece_bytes(h, bit, &bytes, &w), "geometry_piece_bytes");
            Tensor t;
            if (own_pool) {   // (with the geometry's buffer: one pool, one stream to order the memory on)
                const c10::hip::HIPStreamGuardMasqueradingAsCUDA own(as_torch_stream((void*)ss, (int)like.device().index()));
                t = at::empty({(int64_t)(bytes > 256 ? bytes : 256)}, like.options().dtype(at::kByte));
            } else {
                t = at::empty({(int64_t)(bytes > 256 ? bytes : 256)}, like.options().dtype(at::kByte));
            }
            check(mccnn_geometry_attach(h, bit, t.data_ptr(), (size_t)t.numel()), "geometry_attach");
            attached.push_back(std::move(t));
            have |= bit;
            if ((size_t)w > wsb) wsb = (size_t)w;
        }
        Tensor& ws = scratch(wsb, like, (void*)ss);
        // the side stream starts behind the calling stream (the memory just allocated may have had readers there) and,
        // being the build's own stream, behind the geometry itself
        if (!own_pool) {
            static thread_local hipEvent_t fork_ev = nullptr;
            if (!fork_ev) hip_check(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventRecord(fork_ev, (hipStream_t)main_stream), "hipEventRecord");
            hip_check(hipStreamWaitEvent(ss, fork_ev, 0), "hipStreamWaitEvent");
        }
        if (event && side < 0) hip_check(hipStreamWaitEvent(ss, event, 0), "hipStreamWaitEvent");
        check(issue_pieces(what & 7, avg, ss, ws.data_ptr(), (size_t)ws.numel()), "geometry_prebuild");
    }

    int edges(int wait_us) {
        wait_issued();
        if (e < 0 && h) {
            const int v = mccnn_geometry_edges(h, wait_us);
            if (v >= 0) e = v;
        }
        return e;
    }
    std::vector<int64_t> info() {
        wait_issued();
        long long out[16];
        check(mccnn_geometry_info(h, out), "geometry_info");
        return std::vector<int64_t>(out, out + 16);
    }
};

void check_dev(const Tensor& t, at::ScalarType dt, const char* name) {
    TORCH_CHECK(t.defined() && t.is_cuda() && t.scalar_type() == dt && t.is_contiguous(), name,
                ": expected a contiguous device tensor of the op's type");
}

struct HierFuture;
// the event recorded behind an ADOPTED prefetched hierarchy (nullptr: not adopted, failed, none)
hipEvent_t hierarchy_ready_event(const std::shared_ptr<HierFuture>& f);
// whether `t` lives in the memory of that hierarchy (its input points / batch ids, boxes, level rows)
bool hierarchy_owns(const std::shared_ptr<HierFuture>& f, const Tensor& t);

// ---- geometries of a step issued as ONE batch (mccnn_geometry_build_batch: one launch per kernel kind over all of them).
// Between begin_geometry_batch() and end_geometry_batch() every build_geometry() call that would go to a side stream is
// only RECORDED (its buffers allocated, its Geo returned); end_geometry_batch() hands the whole list to the helper thread,
// which issues them on ONE side stream and records every geometry's event behind the batch.
struct BatchEntry {
    std::shared_ptr<Geo> g, grid_from;
    mccnn_geometry_request req;
};
struct PieceEntry {   // row plans / transposed list of a batched geometry whose list is small: built as a batch as well
    std::shared_ptr<Geo> g;
    int what;
    bool avg;
    char* base;
    long long off[4], len[4];
};
struct GeoBatch {
    bool active = false;
    int side = -1;
    bool background = false;
    std::vector<BatchEntry> entries;
    std::vector<PieceEntry> pieces;
    // The buffers of a batch's geometries ([0]) and of their pieces ([1]) are views of ONE allocation each, sized by what the
    // last batch took: a block that carries record_stream() costs its consumer's QUEUE an event record when it is freed
    // (~3-5 us of queue time), and a step's geometries are all joined by the layers' stream and all die in one reset() --
    // 28 blocks per step of BASELINE cfg4 were 0.14 ms of the queue that bounds the step; two blocks are two records.
    Tensor arena[2];
    int64_t arena_off[2] = {0, 0}, taken[2] = {0, 0};
};
thread_local GeoBatch t_geo_batch;
thread_local int64_t t_arena_want[2] = {0, 0};   // bytes the last batch of this thread took from each arena (0: none yet)
// `bytes` of the batch's arena `k` (allocated on first use under the caller's stream guard), or an undefined tensor: the
// caller then allocates a block of its own (first batch, a batch larger than the last one, batches switched off)
Tensor arena_take(int k, int64_t bytes, const Tensor& like) {
    static const bool on = mccnn::debug_int("geo_arena", 1) != 0;   // A/B switch
    GeoBatch& b = t_geo_batch;
    const int64_t need = (bytes + 255) / 256 * 256;
    b.taken[k] += need;
    if (!on || !b.active || t_arena_want[k] <= 0) return Tensor();
    if (!b.arena[k].defined()) {
        b.arena[k] = at::empty({t_arena_want[k] + t_arena_want[k] / 16 + 4096}, like.options().dtype(at::kByte));
        b.arena_off[k] = 0;
    }
    if (b.arena_off[k] + need > b.arena[k].numel()) return Tensor();
    Tensor t = b.arena[k].narrow(0, b.arena_off[k], bytes);
    b.arena_off[k] += need;
    return t;
}
void begin_geometry_batch() {
    static const bool on = mccnn::debug_int("geo_batch", 1) != 0;   // A/B switch: 0 = every geometry its own chain
    t_geo_batch.active = on && Issuer::enabled();
    t_geo_batch.side = -1;
    t_geo_batch.entries.clear();
    t_geo_batch.pieces.clear();
    for (int k = 0; k < 2; ++k) { t_geo_batch.arena[k] = Tensor(); t_geo_batch.arena_off[k] = t_geo_batch.taken[k] = 0; }
}
void end_geometry_batch() {
    GeoBatch& b = t_geo_batch;
    b.active = false;
    for (int k = 0; k < 2; ++k) {   // (the views keep the arenas alive; the next batch is sized by this one)
        if (b.taken[k] > 0) t_arena_want[k] = b.taken[k];
        b.arena[k] = Tensor();
    }
    if (b.entries.empty()) return;
    auto entries = std::make_shared<std::vector<BatchEntry>>(std::move(b.entries));
    b.entries.clear();
    hipStream_t stream = side_stream(b.side);
    const bool background = b.background;
    const int dev = (int)(*entries)[0].g->buf.device().index();
    Issuer::get().push([entries, stream, background, dev] {
        enter_device(dev);
        std::vector<mccnn_geometry_request> reqs;
        reqs.reserve(entries->size());
        for (BatchEntry& e : *entries) {
            // a grid owner outside this batch has to be issued; one inside it is set up by the same library call
            bool inside = false;
            if (e.grid_from)
                for (BatchEntry& o : *entries) inside = inside || (o.g.get() == e.grid_from.get());
            if (e.grid_from && !inside) e.grid_from->wait_issued_nothrow();
            reqs.push_back(e.req);
        }
        const int prev = background ? mccnn_background_launches(1) : 0;
        int rc = mccnn_geometry_build_batch(reqs.data(), (int)reqs.size(), (void*)stream);
        if (background) mccnn_background_launches(prev);
        for (BatchEntry& e : *entries) {
            int r = rc;
            if (r == 0 && hipEventRecord(e.g->event, stream) != hipSuccess) r = (int)hipErrorUnknown;
            e.g->build_rc = r;
            e.g->issued.store(1, std::memory_order_release);
        }
    });
    if (b.pieces.empty()) return;
    // ... and the pieces of the small lists among them: ONE job, one launch per kernel kind (mccnn_geometry_prebuild_batch)
    auto pieces = std::make_shared<std::vector<PieceEntry>>(std::move(b.pieces));
    b.pieces.clear();
    const int pside = (*pieces)[0].g->plan_side;
    Issuer::get(2).push([pieces, pside, dev]() mutable {
        enter_device(dev);
        hipStream_t ss = side_stream(pside);
        std::vector<mccnn_geometry_t*> hs;
        std::vector<int> whats;
        std::vector<PieceEntry*> live;
        bool avg = (*pieces)[0].avg;
        hipEvent_t last = nullptr;
        for (PieceEntry& pe : *pieces) {
            Geo& g = *pe.g;
            g.wait_build_issued_nothrow();
            if (g.build_rc != 0) continue;
            const int prev = mccnn_debug_wait_accounting(0);
            const int E = mccnn_geometry_edges(g.h, -1);
            mccnn_debug_wait_accounting(prev);
            if (E >= 0) g.e.store(E, std::memory_order_relaxed);
            if (E <= 0 || E > g.e_cap) continue;
            int rc = 0;
            for (int k = 0; k < 4 && !rc; ++k)
                if (pe.what & (1 << k)) rc = mccnn_geometry_attach(g.h, 1 << k, pe.base + pe.off[k], (size_t)pe.len[k]);
            if (rc) continue;   // (left unbuilt: the layer that needs a piece builds it, and reports)
            g.have |= pe.what;
            hs.push_back(g.h);
            whats.push_back(pe.what & 7);
            live.push_back(&pe);
            if (g.event) last = g.event;
        }
        int rc = 0;
        if (!hs.empty()) {
            // behind the builds (one stream, events recorded in order: the last one covers them all)
            if (last && hipStreamWaitEvent(ss, last, 0) != hipSuccess) rc = (int)hipErrorUnknown;
            if (!rc) {
                try {
                    const size_t wsb = mccnn_geometry_prebuild_batch_ws_bytes(hs.data(), whats.data(), (int)hs.size());
                    Tensor& ws = scratch(wsb, live[0]->g->buf, (void*)ss);
                    rc = mccnn_geometry_prebuild_batch(hs.data(), whats.data(), (int)hs.size(), avg ? 1 : 0, ws.data_ptr(), (size_t)ws.numel(), (void*)ss);
                } catch (const std::exception&) {
                    rc = (int)hipErrorUnknown;
                }
            }
            if (rc) fprintf(stderr, "mccnn: batch of row plans / transposed lists: error %d (the layers build what is missing)\n", rc);
            // (events also after an error: part of the batch may have been launched, and whoever frees a geometry's memory
            // orders its stream behind them -- Geo::~Geo)
            for (PieceEntry* pe : live) {
                Geo& g = *pe->g;
                if (pe->what & 1) {
                    if (!g.plan_event) g.plan_event = take_event();
                    if (hipEventRecord(g.plan_event, ss) == hipSuccess) g.plan_wait = true;
                }
                if (pe->what & 6) {
                    if (!g.tr_event) g.tr_event = take_event();
                    if (hipEventRecord(g.tr_event, ss) == hipSuccess) g.tr_wait = true;
                }
                g.pre_avg = avg ? 1 : 0;
            }
        }
        for (PieceEntry& pe : *pieces) pe.g->pieces_issued.store(1, std::memory_order_release);
    });
}

std::shared_ptr<Geo> build_geometry(const Tensor& pts, const Tensor& bids, const Tensor& centres, const Tensor& cbids,
                                    const Tensor& mn, const Tensor& mx, int64_t B, int64_t nc, double radius, bool scale_inv,
                                    double window, bool use_pdf, int64_t capacity, std::shared_ptr<Geo> grid_from,
                                    int64_t side, bool fork, bool background, std::shared_ptr<HierFuture> after) {
    check_dev(pts, at::kFloat, "points");
    check_dev(centres, at::kFloat, "sample points");
    check_dev(bids, at::kInt, "batch ids");
    check_dev(cbids, at::kInt, "sample batch ids");
    check_dev(mn, at::kFloat, "aabb_min");
    check_dev(mx, at::kFloat, "aabb_max");
    const DevGuard device_guard((int)pts.device().index());
    const int n = (int)pts.size(0), m = (int)centres.size(0);
    if (grid_from && grid_from->grid_owner) grid_from = grid_from->grid_owner;
    const size_t bytes = mccnn_geometry_bytes(n, m, (int)B, (int)nc, (int)capacity, grid_from ? 0 : 1);
    TORCH_CHECK(bytes > 0, "geometry: batch_size * num_cells^3 does not fit 32-bit keys");
    auto g = std::make_shared<Geo>();
    g->h = mccnn_geometry_create();
    TORCH_CHECK(g->h, "mccnn_geometry_create failed");
    g->slot = take_slot();
    g->keep = {pts, bids, centres, cbids, mn, mx};
    g->grid_owner = grid_from;
    g->n = n; g->m = m; g->nc = (int)nc; g->B = (int)B; g->e_cap = capacity;
    void* stream = cur_stream(pts);
    g->alloc_stream = stream;
    // a geometry that shares another one's grid runs behind it on the same side stream
    if (side >= 0 && grid_from && grid_from->side >= 0 && grid_from->needs_wait) side = grid_from->side;
    // ... and the geometries of a batch all run on the side stream of the first one
    const bool batched = side >= 0 && t_geo_batch.active;
    const int asked_side = side >= 0 ? (int)(((side % kSideStreams) + kSideStreams) % kSideStreams) : -1;
    if (batched) {
        if (t_geo_batch.side < 0) { t_geo_batch.side = (int)(((side % kSideStreams) + kSideStreams) % kSideStreams); t_geo_batch.background = background; }
        side = t_geo_batch.side;
    }
    // Everything this build reads is the adopted prefetched hierarchy `after` (its points, batch ids, boxes -- and a grid
    // that was itself built this way): the build then waits for THAT, takes its memory from its own stream's pool and
    // starts now, not behind what the calling stream holds (Geo::own_pool).
    static const bool own_pools = mccnn::debug_int("geo_own_pool", 1) != 0;
    hipEvent_t ready = (side >= 0 && own_pools && Issuer::enabled()) ? hierarchy_ready_event(after) : nullptr;
    if (ready && grid_from && !grid_from->own_pool) ready = nullptr;
    // ... and only then: an input that is NOT the hierarchy's own memory (a re-made-contiguous copy of a level, a tensor
    // the caller produced on its stream) is ordered by the calling stream alone -- such a build forks behind it like any other
    if (ready)
        for (const Tensor* t : {&pts, &bids, &centres, &cbids, &mn, &mx})
            if (!hierarchy_owns(after, *t)) { ready = nullptr; break; }
    // (a build that takes the fork path after one that did not: the caller's fork=false refers to a record that was skipped)
    static thread_local bool fork_skipped = false;
    if (ready && side_stream((int)side)) {
        hipStream_t ss = side_stream((int)side);
        if (fork) fork_skipped = true;
        {
            const c10::hip::HIPStreamGuardMasqueradingAsCUDA own(as_torch_stream((void*)ss, (int)pts.device().index()));
            if (batched) g->buf = arena_take(0, (int64_t)bytes, pts);
            if (!g->buf.defined()) g->buf = at::empty({(int64_t)bytes}, pts.options().dtype(at::kByte));
        }
        g->own_pool = true;
        g->caller_stream = stream;
        g->has_caller = true;
        g->alloc_stream = (void*)ss;
        hip_check(hipStreamWaitEvent(ss, ready, 0), "hipStreamWaitEvent");
        const int sk = (int)(((side % kSideStreams) + kSideStreams) % kSideStreams);
        if (grid_from && grid_from->event && grid_from->side != sk) {
            grid_from->wait_issued();
            hip_check(hipStreamWaitEvent(ss, grid_from->event, 0), "hipStreamWaitEvent");
        }
        g->side = sk;
        stream = ss;
    } else if (side >= 0) {
        g->buf = at::empty({(int64_t)bytes}, pts.options().dtype(at::kByte));
        hipStream_t ss = side_stream((int)side);
        if (ss) {
            // the side stream starts behind everything the calling stream held at the last fork (the point hierarchy;
            // whatever used the memory the allocator hands out from there on). Only the stream that gets work waits:
            // a wait packet on an idle queue keeps that queue in the command processor's rotation for nothing
            static thread_local hipEvent_t fork_ev = nullptr;
            if (!fork_ev) {
                hip_check(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming), "hipEventCreate");
                fork = true;
            }
            if (fork || fork_skipped) hip_check(hipEventRecord(fork_ev, (hipStream_t)stream), "hipEventRecord");
            fork_skipped = false;
            hip_check(hipStreamWaitEvent(ss, fork_ev, 0), "hipStreamWaitEvent");
            // (a grid owner on another side stream -- its build has been joined by the caller's stream already, or it
            // would have pulled this build onto its own stream above: order this stream behind it without consuming the
            // owner's one-time join)
            if (grid_from && grid_from->event && grid_from->side != (int)(((side % kSideStreams) + kSideStreams) % kSideStreams)) {
                grid_from->wait_issued();
                hip_check(hipStreamWaitEvent(ss, grid_from->event, 0), "hipStreamWaitEvent");
            }
            g->side = (int)(((side % kSideStreams) + kSideStreams) % kSideStreams);
            stream = ss;
        }
    } else {
        g->buf = at::empty({(int64_t)bytes}, pts.options().dtype(at::kByte));
        if (grid_from) grid_from->join(stream);
    }
    if (g->side >= 0 && Issuer::enabled()) {
        // the helper thread issues the launches and records the event; whoever touches the geometry waits for `issued`
        g->event = take_event();
        g->needs_wait = true;
        g->issued.store(0, std::memory_order_release);
        const float* p0 = pts.data_ptr<float>();
        const int* p1 = bids.data_ptr<int>();
        const float* p2 = centres.data_ptr<float>();
        const int* p3 = cbids.data_ptr<int>();
        const float* p4 = mn.data_ptr<float>();
        const float* p5 = mx.data_ptr<float>();
        void* bufp = g->buf.data_ptr();
        int* slotp = g->slot.data_ptr<int>();
        const int iB = (int)B, inc = (int)nc, icap = (int)capacity, isi = scale_inv ? 1 : 0, ipdf = use_pdf ? 1 : 0;
        const float fr = (float)radius, fw = (float)window;
        const int dev = (int)pts.device().index();
        if (batched && stream == (void*)side_stream(t_geo_batch.side)) {   // recorded: end_geometry_batch() issues the list
            BatchEntry e;
            e.g = g;
            e.grid_from = grid_from;
            e.req = mccnn_geometry_request{g->h, p0, p1, n, p2, p3, m, p4, p5, iB, inc, fr, isi, fw, ipdf, icap,
                                           grid_from ? grid_from->h : nullptr, bufp, bytes, slotp};
            t_geo_batch.entries.push_back(std::move(e));
            g->plan_side = asked_side;   // the pieces built ahead keep the spread over the side streams the builds gave up
            return g;
        }
        Issuer::get().push([g, grid_from, p0, p1, p2, p3, p4, p5, bufp, slotp, n, m, iB, inc, icap, isi, ipdf, fr, fw, bytes, stream, background, dev] {
            enter_device(dev);
            if (grid_from) grid_from->wait_issued_nothrow();
            // background: these launches run beside kernels a step waits for (the convolutions of the current batch) --
            // the search kernels then hold back (mccnn_background_launches, thread-local: set on THIS thread)
            const int prev = background ? mccnn_background_launches(1) : 0;
            int rc = mccnn_geometry_build(g->h, p0, p1, n, p2, p3, m, p4, p5, iB, inc, fr, isi, fw, ipdf, icap,
                                          grid_from ? grid_from->h : nullptr, bufp, bytes, slotp, stream);
            if (background) mccnn_background_launches(prev);
            if (rc == 0 && hipEventRecord(g->event, (hipStream_t)stream) != hipSuccess) rc = (int)hipErrorUnknown;
            g->build_rc = rc;
            g->issued.store(1, std::memory_order_release);
        });
        return g;
    }
    if (grid_from) grid_from->wait_issued();
    const int prev_bg = (background && g->side >= 0) ? mccnn_background_launches(1) : 0;
    struct Restore {
        bool on; int prev;
        ~Restore() { if (on) mccnn_background_launches(prev); }
    } restore{background && g->side >= 0, prev_bg};
    check(mccnn_geometry_build(g->h, pts.data_ptr<float>(), bids.data_ptr<int>(), n, centres.data_ptr<float>(),
                               cbids.data_ptr<int>(), m, mn.data_ptr<float>(), mx.data_ptr<float>(), (int)B, (int)nc,
                               (float)radius, scale_inv ? 1 : 0, (float)window, use_pdf ? 1 : 0, (int)capacity,
                               grid_from ? grid_from->h : nullptr, g->buf.data_ptr(), bytes, g->slot.data_ptr<int>(),
                               stream),
          "geometry_build");
    if (g->side >= 0) {
        g->event = take_event();
        hip_check(hipEventRecord(g->event, (hipStream_t)stream), "hipEventRecord");
        g->needs_wait = true;
    }
    return g;
}

// Pieces of a geometry whose build has just been queued on a side stream (learned prefetch: the layers of the previous
// step over this list used them): the buffers are allocated HERE -- on the calling thread and its stream, sized by bounds
// that hold for any list within the capacity -- and the helper thread attaches and issues them behind the build once the
// edge total has arrived. The calling thread neither waits for the total nor issues the ~20 launches of a list's plans.
void prebuild_async(std::shared_ptr<Geo> g, int what, bool avg) {
    if (!g || g->side < 0 || !Issuer::enabled() || g->e_cap <= 0) return;
    const DevGuard device_guard((int)g->buf.device().index());
    if (what & 3) what |= 8;
    if (what & 2) what |= 4;
    what &= 15 & ~g->have;
    if (!what) return;
    long long off[4] = {0, 0, 0, 0}, len[4] = {0, 0, 0, 0}, total = 0, wsb = 256;
    for (int k = 0; k < 4; ++k) {
        if (!(what & (1 << k))) continue;
        long long b = 0, w = 0;
        check(mccnn_geometry_piece_bound(g->n, g->m, (int)g->e_cap, 1 << k, &b, &w), "geometry_piece_bound");
        off[k] = total;
        len[k] = b;
        total += (b + 255) / 256 * 256;
        if (w > wsb) wsb = w;
    }
    // a batched geometry with a small plan (either direction; the other one follows on the same stream): its pieces join the
    // batch too, on ONE side stream (the next one after the builds')
    static const int batch_all = mccnn::debug_int("plan_batch_all", 1);   // A/B: 0 = only geometries with a small plan join the batch
    const bool small_batch = t_geo_batch.active && t_geo_batch.side >= 0 && g->plan_side >= 0 &&
                             (batch_all || mccnn_rowplan_inline_records(g->m, (int)g->e_cap) || mccnn_rowplan_inline_records(g->n, (int)g->e_cap));
    if (small_batch) g->plan_side = (t_geo_batch.side + 1) % kSideStreams;
    const bool other = g->plan_side >= 0 && g->plan_side != g->side;   // (a batch build: pieces on another side stream, behind the build's event)
    hipStream_t ss = side_stream(other ? g->plan_side : g->side);
    Tensor block;
    if (g->own_pool) {
        const c10::hip::HIPStreamGuardMasqueradingAsCUDA own(as_torch_stream((void*)ss, (int)g->buf.device().index()));
        if (small_batch) block = arena_take(1, (int64_t)total, g->buf);
        if (!block.defined()) block = at::empty({(int64_t)total}, g->buf.options());
    } else {
        block = at::empty({(int64_t)total}, g->buf.options());
    }
    g->attached.push_back(block);
    g->pieces_issued.store(0, std::memory_order_release);
    char* base = (char*)block.data_ptr();
    Tensor like = g->buf;
    if (small_batch) {
        PieceEntry pe;
        pe.g = g; pe.what = what; pe.avg = avg; pe.base = base;
        for (int k = 0; k < 4; ++k) { pe.off[k] = off[k]; pe.len[k] = len[k]; }
        t_geo_batch.pieces.push_back(std::move(pe));
        return;
    }
    Issuer::get(2).push([g, what, avg, base, off, len, wsb, ss, like, other]() mutable {
        enter_device((int)like.device().index());
        g->wait_build_issued_nothrow();
        if (g->build_rc == 0 && other && g->event && hipStreamWaitEvent(ss, g->event, 0) != hipSuccess) g->build_rc = (int)hipErrorUnknown;
        if (g->build_rc == 0) {
            const int prev = mccnn_debug_wait_accounting(0);
            const int E = mccnn_geometry_edges(g->h, -1);
            mccnn_debug_wait_accounting(prev);
            if (E >= 0) g->e.store(E, std::memory_order_relaxed);
            if (E > 0 && E <= g->e_cap) {
                int rc = 0;
                for (int k = 0; k < 4 && !rc; ++k)
                    if (what & (1 << k)) rc = mccnn_geometry_attach(g->h, 1 << k, base + off[k], (size_t)len[k]);
                if (!rc) {
                    g->have |= what;
                    try {
                        Tensor& ws = scratch((size_t)wsb, like, (void*)ss);
                        rc = g->issue_pieces(what & 7, avg, ss, ws.data_ptr(), (size_t)ws.numel());
                    } catch (const std::exception&) {
                        rc = (int)hipErrorUnknown;
                    }
                }
                // (a failure here leaves the pieces attached but unbuilt: the layer that needs one builds it, and reports)
            }
        }
        g->pieces_issued.store(1, std::memory_order_release);
    });
}

struct Layer {
    int fin, fout, combin, avg, bf16, flags;
};

// mccnn_conv_prepare + the attachments it asks for; -> scratch bytes, saved bytes
void prepare(Geo& g, const Tensor& feats, const Layer& L, int backward, int flags, long long& ws_bytes, long long& saved_bytes) {
    int mask = 0, edges = 0;
    long long need[4] = {0, 0, 0, 0};
    int rc = mccnn_conv_prepare(g.h, feats.data_ptr(), L.fin, L.fout, L.combin, L.bf16, backward, flags, &mask, need, &ws_bytes,
                                &saved_bytes, &edges);
    if (rc == MCCNN_E_CAPACITY) {
        g.e = edges;
        throw CapacityError(edges);
    }
    check(rc, "conv_prepare");
    g.e = edges;
    for (int k = 0; k < 4 && mask; ++k) {
        const int bit = 1 << k;
        if (!(mask & bit)) continue;
        Tensor t = at::empty({(int64_t)(need[k] > 256 ? need[k] : 256)}, feats.options().dtype(at::kByte));
        check(mccnn_geometry_attach(g.h, bit, t.data_ptr(), (size_t)t.numel()), "geometry_attach");
        g.attached.push_back(std::move(t));
        g.have |= bit;
    }
}

// host-time census of the layer calls (debug_times(): ns and calls per section; cheap enough to stay compiled in)
enum { T_FWD = 0, T_FWD_LIB, T_FWD_ALLOC, T_BWD, T_BWD_LIB, T_BWD_ALLOC, T_BWD_VIEWS, T_BWD_JOIN, T_N };
std::atomic<long long> g_t_ns[T_N], g_t_calls[T_N];
struct Tick {
    int k;
    std::chrono::steady_clock::time_point t0;
    explicit Tick(int k_) : k(k_), t0(std::chrono::steady_clock::now()) {}
    ~Tick() {
        g_t_ns[k].fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
        g_t_calls[k].fetch_add(1, std::memory_order_relaxed);
    }
};

struct ConvBackward : public torch::autograd::Node {
    std::shared_ptr<Geo> geo;
    // the seven inputs, held as plain tensors with the version each had in the forward pass (they are INPUTS of this node,
    // so holding them makes no cycle; SavedVariable::unpack builds a fresh Variable per tensor and call, ~5 us per node --
    // what it guards against, an input modified in place since the forward pass, is checked here directly)
    Tensor feats_, w1_, b1_, w2_, b2_, w3_, b3_;
    uint32_t versions[7] = {0, 0, 0, 0, 0, 0, 0};
    Tensor saved;
    Layer L;

    torch::autograd::variable_list apply(torch::autograd::variable_list&& grads) override {
        const Tick tick_all(T_BWD);
        TORCH_CHECK(geo && feats_.defined(), "MC convolution: backward through a graph whose buffers have been freed (retain_graph=True?)");
        const DevGuard device_guard((int)geo->buf.device().index());
        const Tensor &feats = feats_, &w1 = w1_, &b1 = b1_, &w2 = w2_, &b2 = b2_, &w3 = w3_, &b3 = b3_;
        {
            const Tensor* in[7] = {&feats_, &w1_, &b1_, &w2_, &b2_, &w3_, &b3_};
            for (int k = 0; k < 7; ++k)
                TORCH_CHECK(in[k]->_version() == versions[k],
                            "one of the variables needed for gradient computation has been modified by an inplace operation "
                            "(MC convolution input ", k, ")");
        }
        Tensor og = grads[0];
        if (!og.defined()) og = at::zeros({geo->m, L.combin ? L.fout : L.fin}, feats.options());
        if (!og.is_contiguous()) og = og.contiguous();
        if (og.scalar_type() != feats.scalar_type()) og = og.to(feats.scalar_type());
        int flags = L.flags;
        // several layers share this neighbour list: its transposed form is built once and the feature gradient of combin
        // layers with 2..4 input features is gathered through it in a fixed order (bit-reproducible) instead of added
        // with float atomics; a bare single call keeps the atomics (the list would cost more than they do)
        if (geo->uses > 1) flags |= 2;
        { const Tick tj(T_BWD_JOIN); geo->join(cur_stream(feats), true); }
        long long wsb = 0, svb = 0;
        prepare(*geo, feats, L, 1, flags, wsb, svb);
        Tensor fg;
        Tensor gflat;
        // the six MLP gradients: consecutive slices of ONE buffer in the order the builder creates the variables (a
        // data-parallel step all-reduces that buffer as it is, dist.GradBucket)
        const int64_t n1 = w1.numel(), n2 = b1.numel(), n3 = w2.numel(), n4 = b2.numel(), n5 = w3.numel(), n6 = b3.numel();
        {
            const Tick ta(T_BWD_ALLOC);
            fg = at::empty_like(feats);
            gflat = at::empty({n1 + n2 + n3 + n4 + n5 + n6}, w1.options());
        }
        float* base = gflat.data_ptr<float>();
        void* st = cur_stream(feats);
        Tensor& ws = scratch((size_t)wsb, feats, st);
        {
        const Tick tl(T_BWD_LIB);
        check(mccnn_conv_backward(geo->h, feats.data_ptr(), saved.defined() ? saved.data_ptr() : nullptr,
                                  saved.defined() ? (size_t)saved.numel() : 0, og.data_ptr(), L.fin, L.fout, L.combin, L.avg,
                                  L.bf16, flags, w1.data_ptr<float>(), b1.data_ptr<float>(), w2.data_ptr<float>(),
                                  b2.data_ptr<float>(), w3.data_ptr<float>(), b3.data_ptr<float>(), fg.data_ptr(), base,
                                  base + n1, base + n1 + n2, base + n1 + n2 + n3, base + n1 + n2 + n3 + n4,
                                  base + n1 + n2 + n3 + n4 + n5, ws.data_ptr(), (size_t)ws.numel(), st),
              "conv_backward");
        }
        const Tick tv(T_BWD_VIEWS);
        int64_t o = 0;
        auto piece = [&](int64_t cnt, const Tensor& like) {
            Tensor t = gflat.as_strided(like.sizes(), like.strides(), o);   // (contiguous variables: checked in conv())
            o += cnt;
            return t;
        };
        torch::autograd::variable_list out(7);
        out[0] = fg;
        out[1] = piece(n1, w1); out[2] = piece(n2, b1); out[3] = piece(n3, w2); out[4] = piece(n4, b2);
        out[5] = piece(n5, w3); out[6] = piece(n6, b3);
        return out;
    }

    void release_variables() override {
        feats_.reset(); w1_.reset(); b1_.reset(); w2_.reset(); b2_.reset(); w3_.reset(); b3_.reset();
        saved.reset();
        geo.reset();
    }
};

// One MC convolution over `geo` (SpatialConv with sort_features folded in, MCConvModuleSrc:35-45,70-81): feats are the
// rows of the UNSORTED input points ([n, Fin] f32, or bf16 for depth-wise layers); the kernel-MLP tensors in any shape
// over the reference's flat layout.
Tensor conv(std::shared_ptr<Geo> geo, const Tensor& feats, const Tensor& w1, const Tensor& b1, const Tensor& w2,
            const Tensor& b2, const Tensor& w3, const Tensor& b3, int64_t fout, bool combin, bool avg, bool deterministic) {
    TORCH_CHECK(geo && geo->h, "conv: no geometry");
    const DevGuard device_guard((int)feats.device().index());
    TORCH_CHECK(feats.defined() && feats.is_cuda() && feats.dim() == 2 && feats.size(0) == geo->n && feats.is_contiguous(),
                "SpatialConvOp expects as feature inputs the following dimensions (numPoints, numFeatures)");
    const bool bf = feats.scalar_type() == at::kBFloat16;
    TORCH_CHECK(bf || feats.scalar_type() == at::kFloat, "features must be float32 or bfloat16");
    for (const Tensor* t : {&w1, &b1, &w2, &b2, &w3, &b3}) check_dev(*t, at::kFloat, "kernel-MLP tensor");
    Layer L;
    L.fin = (int)feats.size(1); L.fout = (int)fout; L.combin = combin ? 1 : 0; L.avg = avg ? 1 : 0; L.bf16 = bf ? 1 : 0;
    const bool need_grad = at::GradMode::is_enabled() &&
                           (feats.requires_grad() || w1.requires_grad() || b1.requires_grad() || w2.requires_grad() ||
                            b2.requires_grad() || w3.requires_grad() || b3.requires_grad());
    // bit 1: the caller asks for bit-reproducible feature gradients (combin layers with 2..4 input features gather them
    // through the transposed list instead of adding them with float atomics)
    L.flags = (need_grad ? 1 : 0) | (deterministic ? 2 : 0);
    Tensor out, saved;
    const Tick tick_all(T_FWD);
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        geo->join(cur_stream(feats), geo->pre_avg != L.avg);
        if (geo->grid_owner) geo->grid_owner->join(cur_stream(feats));
        long long wsb = 0, svb = 0;
        prepare(*geo, feats, L, 0, L.flags, wsb, svb);
        {
            const Tick ta(T_FWD_ALLOC);
            out = at::empty({geo->m, combin ? fout : (int64_t)L.fin}, feats.options());
            if (svb > 0) saved = at::empty({(int64_t)svb}, feats.options().dtype(at::kByte));
        }
        void* st = cur_stream(feats);
        Tensor& ws = scratch((size_t)wsb, feats, st);
        const Tick tl(T_FWD_LIB);
        check(mccnn_conv_forward(geo->h, feats.data_ptr(), L.fin, L.fout, L.combin, L.avg, L.bf16, L.flags, w1.data_ptr<float>(),
                                 b1.data_ptr<float>(), w2.data_ptr<float>(), b2.data_ptr<float>(), w3.data_ptr<float>(),
                                 b3.data_ptr<float>(), out.data_ptr(), saved.defined() ? saved.data_ptr() : nullptr,
                                 (size_t)svb, ws.data_ptr(), (size_t)ws.numel(), st),
              "conv_forward");
    }
    if (need_grad) {
        auto node = std::shared_ptr<ConvBackward>(new ConvBackward(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(feats, w1, b1, w2, b2, w3, b3));
        node->geo = geo;
        node->feats_ = feats; node->w1_ = w1; node->b1_ = b1; node->w2_ = w2; node->b2_ = b2; node->w3_ = w3; node->b3_ = b3;
        {
            const Tensor* in[7] = {&feats, &w1, &b1, &w2, &b2, &w3, &b3};
            for (int k = 0; k < 7; ++k) node->versions[k] = in[k]->_version();
        }
        node->saved = saved;
        node->L = L;
        torch::autograd::set_history(out, node);
    }
    return out;
}

// ComputeAabb through the extension (aabb_gpu.cc:22-86): the op a hierarchy without a prefetch request starts with, every
// step -- one C++ call instead of the ctypes op (two allocations, a workspace look-up, ten argument conversions).
std::vector<Tensor> compute_aabb(const Tensor& pts, const Tensor& bids, int64_t B, bool scale_inv) {
    check_dev(pts, at::kFloat, "points");
    check_dev(bids, at::kInt, "batch ids");
    TORCH_CHECK(B > 0, "ComputeAabbOp expects a positive batch size");
    const DevGuard device_guard((int)pts.device().index());
    Tensor mn = at::empty({B, 3}, pts.options()), mx = at::empty({B, 3}, pts.options());
    void* st = cur_stream(pts);
    Tensor& ws = scratch(mccnn_compute_aabb_workspace_bytes((int)B), pts, st);
    check(mccnn_compute_aabb(pts.data_ptr<float>(), bids.data_ptr<int>(), (int)pts.size(0), (int)B, scale_inv ? 1 : 0, mn.data_ptr<float>(),
                             mx.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(), st),
          "compute_aabb");
    return {mn, mx};
}

std::atomic<long long> g_wait_ns{0};  // host time spent waiting for the level sizes (diagnostics: wait_ns())
void count_wait(std::chrono::steady_clock::time_point t0) {
    if (t_helper_thread) return;   // (only the calling side's waits: a helper thread waiting is the point of having it)
    g_wait_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
                        std::memory_order_relaxed);
}

// Geometry of ALL levels of a point hierarchy (MCConvBuilder.py:101-128 per level: sort_points_step1/2 -> poisson_sampling
// -> transform_indexs) with device-side point counts and ONE read-back of the level sizes at the end: the C++ form of
// MCConvModule.point_hierarchy_levels (one library call per level, two allocations, no Python in between).
// -> per level (sampledPts [S,3], sampledBatchIds [S,1], sampledIndexs [S], transformedIndexs [S]); an empty vector when a
// wait of the single-launch Poisson kernel timed out (the caller then runs the op-by-op chain).
std::vector<std::vector<Tensor>> hierarchy_levels(const Tensor& pts, const Tensor& bids, const Tensor& mn, const Tensor& mx,
                                                  const std::vector<double>& radii, const std::vector<int64_t>& ncs,
                                                  int64_t B, bool scale_inv, int64_t pmode) {
    check_dev(pts, at::kFloat, "points");
    check_dev(bids, at::kInt, "batch ids");
    check_dev(mn, at::kFloat, "aabb_min");
    check_dev(mx, at::kFloat, "aabb_max");
    const DevGuard device_guard((int)pts.device().index());
    const int L = (int)radii.size();
    const int cap = (int)pts.size(0);
    TORCH_CHECK(L > 0 && (int)ncs.size() == L && cap > 0, "hierarchy_levels: bad arguments");
    void* st = cur_stream(pts);
    auto iopt = pts.options().dtype(at::kInt);
    Tensor sizes = at::empty({L + 1}, iopt);
    hip_check(hipMemcpyAsync(sizes.data_ptr(), &cap, sizeof(int), hipMemcpyHostToDevice, (hipStream_t)st), "hipMemcpyAsync");
    const int64_t ca = (cap + 63) / 64 * 64;  // 256-byte aligned pieces (the cell table is written as int2)
    std::vector<Tensor> ints(L), flts(L);
    const float* cur_pts = pts.data_ptr<float>();
    const int* cur_bids = bids.data_ptr<int>();
    int* sz = sizes.data_ptr<int>();
    for (int l = 0; l < L; ++l) {
        const int nc = (int)ncs[l];
        const size_t wsb = mccnn_hierarchy_level_workspace_bytes(cap, (int)B, nc);
        TORCH_CHECK(wsb > 0, "PointHierarchy: batch_size * num_cells^3 does not fit 32-bit keys");
        Tensor& ws = scratch(wsb, pts, st);
        // index_new_pos | sorted batch ids | oB | oI | ti | cell table          and          sorted points | sampled points
        ints[l] = at::empty({5 * ca + 2 * B * nc * nc * nc}, iopt);
        flts[l] = at::empty({6 * ca}, pts.options());
        int* bi = ints[l].data_ptr<int>();
        float* bf = flts[l].data_ptr<float>();
        check(mccnn_hierarchy_level(cur_pts, cur_bids, mn.data_ptr<float>(), mx.data_ptr<float>(), cap, sz + l, (int)B, nc,
                                    (float)radii[l], scale_inv ? 1 : 0, (int)pmode, bi, bf, bi + ca, bi + 5 * ca, bf + 3 * ca,
                                    bi + 2 * ca, bi + 3 * ca, bi + 4 * ca, sz + l + 1, ws.data_ptr(), (size_t)ws.numel(), st),
              "hierarchy_level");
        cur_pts = bf + 3 * ca;
        cur_bids = bi + 2 * ca;
    }
    // the ONE read-back: every level's sample count
    static thread_local Tensor host;
    if (!host.defined() || host.numel() < L + 1) host = at::empty({L + 65}, at::TensorOptions().dtype(at::kInt).pinned_memory(true));
    hip_check(hipMemcpyAsync(host.data_ptr(), sz, (size_t)(L + 1) * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)st), "hipMemcpyAsync");
    {
        const auto t0 = std::chrono::steady_clock::now();
        hip_check(hipStreamSynchronize((hipStream_t)st), "hipStreamSynchronize");
        g_wait_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
                            std::memory_order_relaxed);
    }
    const int* hs = host.data_ptr<int>();
    std::vector<std::vector<Tensor>> out;
    for (int l = 1; l <= L; ++l)
        if (hs[l] < 0) return out;
    for (int l = 0; l < L; ++l) {
        const int64_t sN = hs[l + 1];
        Tensor oP = flts[l].narrow(0, 3 * ca, 3 * sN).view({sN, 3});
        Tensor oB = ints[l].narrow(0, 2 * ca, sN).view({sN, 1});
        Tensor oI = ints[l].narrow(0, 3 * ca, sN);
        Tensor ti = ints[l].narrow(0, 4 * ca, sN);
        out.push_back({oP, oB, oI, ti});
    }
    return out;
}


// ---- GetSampledFeatures (+Grad), MCConvModuleSrc:63-68: out[i, :] = feats[idx[i], :]; the gradient scatters into zeros.
// The C++ form of MCConvModule.get_sampled_features: a hierarchy gathers the feature rows of every level, one call each.
struct GatherBackward : public torch::autograd::Node {
    Tensor idx;
    int64_t n = 0;
    torch::autograd::variable_list apply(torch::autograd::variable_list&& grads) override {
        Tensor g = grads[0];
        TORCH_CHECK(g.defined() && idx.defined(), "GetSampledFeaturesGrad: no gradient / released buffers");
        if (!g.is_contiguous()) g = g.contiguous();
        const DevGuard device_guard((int)g.device().index());
        const int64_t words = g.size(1) * (int64_t)g.element_size() / 4;
        Tensor out = at::empty({n, g.size(1)}, g.options());
        check(mccnn_permute_scatter((const float*)g.data_ptr(), idx.data_ptr<int>(), (int)g.size(0), (int)words,
                                    (float*)out.data_ptr(), (int)n, 1, cur_stream(g)),
              "permute_scatter");
        return {out};
    }
    void release_variables() override { idx.reset(); }
};

Tensor sampled_features(const Tensor& idx, const Tensor& feats) {
    check_dev(idx, at::kInt, "sampled indexs");
    TORCH_CHECK(idx.dim() == 1, "GetSampledFeaturesOp expects indexs with the following dimensions (numSamples)");
    TORCH_CHECK(feats.defined() && feats.is_cuda() && feats.dim() == 2 && feats.size(1) > 0 && feats.is_contiguous() &&
                    (feats.scalar_type() == at::kFloat || (feats.scalar_type() == at::kBFloat16 && feats.size(1) % 2 == 0)),
                "GetSampledFeaturesOp expects features with dimensions (numPoints, numFeatures)");
    const DevGuard device_guard((int)feats.device().index());
    const int64_t words = feats.size(1) * (int64_t)feats.element_size() / 4;
    Tensor out;
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        out = at::empty({idx.size(0), feats.size(1)}, feats.options());
        check(mccnn_permute_gather((const float*)feats.data_ptr(), idx.data_ptr<int>(), (int)idx.size(0), (int)words,
                                   (float*)out.data_ptr(), cur_stream(feats)),
              "permute_gather");
    }
    if (at::GradMode::is_enabled() && feats.requires_grad()) {
        auto node = std::shared_ptr<GatherBackward>(new GatherBackward(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(feats));
        node->idx = idx;
        node->n = feats.size(0);
        torch::autograd::set_history(out, node);
    }
    return out;
}

// ---- point hierarchy of the NEXT batch, built on a stream of its own by a helper thread -------------------------------
// A hierarchy depends on the points only. Built inline it is a chain of ~13 small dependent kernels per level that ends in
// a read-back of the level sizes (and, for an absolute radius, starts with one of the box extent): the host cannot issue
// anything of the step while it waits, and the GPU has nothing else to run. Requested one step ahead the chain runs under
// the convolutions of the batch in flight, and the calling thread never waits for the device.
// Memory: everything the caller will read (boxes, level rows) is allocated HERE, on the calling thread and its current
// stream, sized by the capacity (level 0's point count) -- the helper's stream starts behind an event recorded on the
// caller's stream at this moment, and the caller's stream joins the helper's event before the first use, so the blocks
// go back to the allocator they came from after every reader. Temporaries (sorted copies, cell tables, scan words) live
// in the helper thread's own scratch.
hipStream_t hier_stream() {
    static hipStream_t streams[kMaxDevices] = {};
    static bool made[kMaxDevices] = {};
    static std::mutex m;
    const int d = device_now();
    std::lock_guard<std::mutex> lk(m);
    if (!made[d]) {
        if (hipStreamCreateWithFlags(&streams[d], hipStreamNonBlocking) != hipSuccess) streams[d] = nullptr;
        made[d] = true;
    }
    return streams[d];
}

struct HierFuture {
    Tensor pts, bids, mn, mx, sizes;
    Tensor block;                    // prefetched hierarchies: the ONE allocation mn .. lfeats are views of
    std::vector<Tensor> ints, flts;  // per level: sampled batch ids | sampled indexs | transformed indexs   and   sampled points
    // optional: the input feature rows of level 0 (no gradient, short rows) -- every level's rows are then gathered HERE,
    // on the hierarchy's stream, once the level sizes are known (GetSampledFeatures, MCConvBuilder.py:112-116), instead
    // of by one launch per level on the calling thread when the hierarchy is adopted
    Tensor feats;
    std::vector<Tensor> lfeats;      // per level: [ca x F] rows
    std::vector<double> radii;
    std::vector<int> ncs, hs;
    int B = 0, L = 0, cap = 0, pmode = 1;
    int64_t ca = 0;
    bool scale_inv = true;
    float extent = 0.f;
    std::atomic<int> done{1};
    int rc = 0;
    std::string what;
    hipEvent_t event = nullptr;
    void* alloc_stream = nullptr;
    bool joined = false;

    void wait_done() {
        int spins = 0;
        while (!done.load(std::memory_order_acquire)) {
            if (++spins < 2000) continue;
            if (spins < 4000) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    ~HierFuture() {
        wait_done();
        if (event) {
            if (!joined && rc == 0) (void)hipStreamWaitEvent((hipStream_t)alloc_stream, event, 0);
            give_event(event);
        }
    }

    // the helper thread's part: boxes, cell counts, the levels, ONE read-back of the sizes
    void run(hipStream_t ss) {
        try {
            enter_device((int)pts.device().index());
            const float* P = pts.data_ptr<float>();
            const int* Bi = bids.data_ptr<int>();
            float* pmn = mn.data_ptr<float>();
            float* pmx = mx.data_ptr<float>();
            const size_t aw = mccnn_compute_aabb_workspace_bytes(B);
            check(mccnn_compute_aabb(P, Bi, cap, B, scale_inv ? 1 : 0, pmn, pmx, scratch(aw, pts, (void*)ss).data_ptr(),
                                     (size_t)scratch(aw, pts, (void*)ss).numel(), (void*)ss),
                  "compute_aabb");
            ncs.assign(L, 1);
            if (scale_inv) {
                for (int l = 0; l < L; ++l)
                    check(mccnn_num_cells(nullptr, nullptr, B, (float)radii[l], 1, &ncs[l], nullptr), "num_cells");
            } else {
                // determineNumCells with an absolute cell size (sort_gpu.cu:410-419): ONE read-back of the extent for all levels
                check(mccnn_aabb_extent(pmn, pmx, &extent, (void*)ss), "aabb_extent");
                for (int l = 0; l < L; ++l) {
                    const int nc = (int)(extent / (float)radii[l]);
                    ncs[l] = nc ? nc : 1;
                }
            }
            size_t need = 256;
            std::vector<size_t> wsb(L);
            for (int l = 0; l < L; ++l) {
                wsb[l] = mccnn_hierarchy_level_workspace_bytes(cap, B, ncs[l]);
                if (!wsb[l]) throw std::runtime_error("PointHierarchy: batch_size * num_cells^3 does not fit 32-bit keys");
                wsb[l] = (wsb[l] + 255) / 256 * 256;
                const size_t tmp = (size_t)(5 * ca) * 4 + (size_t)2 * B * ncs[l] * ncs[l] * ncs[l] * 4;
                if (wsb[l] + tmp > need) need = wsb[l] + tmp;
            }
            Tensor& ws = scratch(need, pts, (void*)ss);
            int* sz = sizes.data_ptr<int>();
            hip_check(hipMemcpyAsync(sz, &cap, sizeof(int), hipMemcpyHostToDevice, ss), "hipMemcpyAsync");
            const float* cur_pts = P;
            const int* cur_bids = Bi;
            for (int l = 0; l < L; ++l) {
                char* base = (char*)ws.data_ptr();
                int* ti_ = (int*)(base + wsb[l]);           // index_new_pos | sorted batch ids | cell table
                float* tf_ = (float*)(ti_ + 2 * ca);        // sorted points
                int* cells = (int*)(tf_ + 3 * ca);
                int* bi = ints[l].data_ptr<int>();
                float* bf = flts[l].data_ptr<float>();
                check(mccnn_hierarchy_level(cur_pts, cur_bids, pmn, pmx, cap, sz + l, B, ncs[l], (float)radii[l],
                                            scale_inv ? 1 : 0, pmode, ti_, tf_, ti_ + ca, cells, bf, bi, bi + ca, bi + 2 * ca,
                                            sz + l + 1, base, wsb[l], (void*)ss),
                      "hierarchy_level");
                cur_pts = bf;
                cur_bids = bi;
            }
            static thread_local Tensor host;
            if (!host.defined() || host.numel() < L + 1)
                host = at::empty({L + 65}, at::TensorOptions().dtype(at::kInt).pinned_memory(true));
            hip_check(hipMemcpyAsync(host.data_ptr(), sz, (size_t)(L + 1) * sizeof(int), hipMemcpyDeviceToHost, ss), "hipMemcpyAsync");
            if (!feats.defined()) hip_check(hipEventRecord(event, ss), "hipEventRecord");
            hip_check(hipStreamSynchronize(ss), "hipStreamSynchronize");
            hs.assign(host.data_ptr<int>(), host.data_ptr<int>() + L + 1);
            for (int l = 1; l <= L; ++l)
                if (hs[l] < 0) {   // (rare and worth knowing: a wait of the single-launch Poisson kernel timed out under load)
                    fprintf(stderr, "mccnn: prefetched hierarchy: level %d gave up (size %d); the caller builds it op by op\n", l, hs[l]);
                    break;
                }
            if (feats.defined()) {
                bool ok = true;
                for (int l = 1; l <= L; ++l) ok = ok && hs[l] >= 0;
                const int words = (int)(feats.size(1) * (int64_t)feats.element_size() / 4);
                const float* src = (const float*)feats.data_ptr();
                for (int l = 0; ok && l < L; ++l) {   // rows of level l + 1 = rows of level l at transformedIndexs
                    check(mccnn_permute_gather(src, ints[l].data_ptr<int>() + 2 * ca, hs[l + 1], words,
                                               (float*)lfeats[l].data_ptr(), (void*)ss), "permute_gather");
                    src = (const float*)lfeats[l].data_ptr();
                }
                hip_check(hipEventRecord(event, ss), "hipEventRecord");
            }
            if (mccnn::debug_int("hier_trace", 0)) {
                std::string line = "hier job: cap " + std::to_string(cap) + " extent " + std::to_string(extent) + " nc";
                for (int l = 0; l < L; ++l) line += " " + std::to_string(ncs[l]);
                line += " sizes";
                for (int l = 0; l <= L; ++l) line += " " + std::to_string(hs[l]);
                fprintf(stderr, "%s\n", line.c_str());
            }
        } catch (const std::exception& e) {
            rc = 1;
            what = e.what();
        }
        done.store(1, std::memory_order_release);
    }

    // -> (aabbMin, aabbMax, extent, levels) on the calling thread's current stream; levels empty when a wait of the
    // single-launch Poisson kernel timed out (the caller then builds the hierarchy op by op)
    py::tuple result() {
        {
            py::gil_scoped_release nogil;
            const auto t0 = std::chrono::steady_clock::now();
            wait_done();
            g_wait_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
                                std::memory_order_relaxed);
        }
        if (rc) throw std::runtime_error("PointHierarchy prefetch: " + what);
        const DevGuard device_guard((int)pts.device().index());
        if (!joined) {
            void* consumer_stream = cur_stream(pts);
            hip_check(hipStreamWaitEvent((hipStream_t)consumer_stream, event, 0), "hipStreamWaitEvent");
            joined = true;
            // (allocated on the hierarchy's stream, consumed on this one from here on)
            const c10::Stream consumer = as_torch_stream(consumer_stream, (int)pts.device().index());
            if (block.defined()) {
                block.record_stream(consumer);   // (every output is a view of it)
            } else {
                mn.record_stream(consumer);
                mx.record_stream(consumer);
                for (Tensor& t : ints) t.record_stream(consumer);
                for (Tensor& t : flts) t.record_stream(consumer);
                for (Tensor& t : lfeats) t.record_stream(consumer);
            }
        }
        std::vector<std::vector<Tensor>> out;
        bool ok = true;
        for (int l = 1; l <= L; ++l) ok = ok && hs[l] >= 0;
        for (int l = 0; ok && l < L; ++l) {
            const int64_t sN = hs[l + 1];
            out.push_back({flts[l].narrow(0, 0, 3 * sN).view({sN, 3}), ints[l].narrow(0, 0, sN).view({sN, 1}),
                           ints[l].narrow(0, ca, sN), ints[l].narrow(0, 2 * ca, sN)});
            if (feats.defined()) {   // 5th entry: the level's feature rows
                const int64_t F = feats.size(1);
                out.back().push_back(lfeats[l].narrow(0, 0, sN * F).view({sN, F}));
            }
        }
        return py::make_tuple(mn, mx, extent, out);
    }
};

hipEvent_t hierarchy_ready_event(const std::shared_ptr<HierFuture>& f) {
    return (f && f->joined && f->rc == 0 && f->done.load(std::memory_order_acquire)) ? f->event : nullptr;
}
bool hierarchy_owns(const std::shared_ptr<HierFuture>& f, const Tensor& t) {
    if (!f || !t.defined() || !t.is_cuda()) return false;
    const char* p = static_cast<const char*>(t.data_ptr());
    auto inside = [p](const Tensor& o) {
        if (!o.defined() || !o.is_cuda()) return false;
        const char* b = static_cast<const char*>(o.data_ptr());
        return p >= b && p < b + o.nbytes();
    };
    if (inside(f->pts) || inside(f->bids) || inside(f->mn) || inside(f->mx)) return true;
    for (const Tensor& o : f->ints) if (inside(o)) return true;
    for (const Tensor& o : f->flts) if (inside(o)) return true;
    for (const Tensor& o : f->lfeats) if (inside(o)) return true;
    return false;
}

// after_mode: 0 = the build starts behind everything the calling stream holds now (where the inputs were produced);
// 1 = the inputs are complete (uploaded and synchronised, or resident for long): the build starts at once;
// 2 = behind `after_event` (a hipEvent_t: the upload's own stream recorded it).
std::shared_ptr<HierFuture> hierarchy_prefetch(const Tensor& pts, const Tensor& bids, const std::vector<double>& radii,
                                               int64_t B, bool scale_inv, int64_t pmode, int64_t after_mode,
                                               int64_t after_event, const c10::optional<Tensor>& feats) {
    check_dev(pts, at::kFloat, "points");
    check_dev(bids, at::kInt, "batch ids");
    const DevGuard device_guard((int)pts.device().index());
    const int L = (int)radii.size();
    const int cap = (int)pts.size(0);
    TORCH_CHECK(L > 0 && cap > 0 && B > 0, "hierarchy_prefetch: bad arguments");
    hipStream_t ss = hier_stream();
    TORCH_CHECK(ss, "hierarchy_prefetch: no side stream");
    auto f = std::make_shared<HierFuture>();
    f->pts = pts; f->bids = bids; f->radii = radii;
    f->B = (int)B; f->L = L; f->cap = cap; f->pmode = (int)pmode; f->scale_inv = scale_inv;
    f->ca = (cap + 63) / 64 * 64;
    auto iopt = pts.options().dtype(at::kInt);
    void* stream = cur_stream(pts);
    {
        // the outputs come from the caching allocator's pool of the hierarchy's OWN stream: a block it hands out was last
        // used on that stream, so the build needs no ordering behind the calling stream for its memory (only for its
        // inputs, below); result() tells the allocator about the stream that consumes them (record_stream)
        const c10::hip::HIPStreamGuardMasqueradingAsCUDA own(as_torch_stream((void*)ss, (int)pts.device().index()));
        bool with_feats = false;
        if (feats.has_value() && feats->defined()) {
            const Tensor& ft = *feats;
            // short rows without a gradient only (the input features of a network: ones, normals, colours): the level
            // buffers are sized by the capacity
            const int64_t row_bytes = ft.dim() == 2 ? ft.size(1) * (int64_t)ft.element_size() : 0;
            with_feats = ft.is_cuda() && ft.device() == pts.device() && ft.dim() == 2 && ft.size(0) == cap && ft.is_contiguous() &&
                         !ft.requires_grad() && row_bytes > 0 && row_bytes <= 256 && row_bytes % 4 == 0 &&
                         (ft.scalar_type() == at::kFloat || ft.scalar_type() == at::kBFloat16);
            if (with_feats) f->feats = ft;
        }
        // ONE block for everything the hierarchy hands out (boxes, level sizes, every level's rows): a block that carries
        // record_stream() costs its consumer's QUEUE an event record when it is freed (~3-5 us of queue time each; a dozen
        // blocks per hierarchy were 0.08 ms of the calling queue per step of BASELINE cfg4) -- one block, one record
        auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
        const int64_t bBox = al(B * 3 * 4), bSz = al((L + 1) * 4), bLvl = al(3 * f->ca * 4);
        const int64_t bFt = with_feats ? al(f->ca * f->feats.size(1) * (int64_t)f->feats.element_size()) : 0;
        f->block = at::empty({2 * bBox + bSz + L * (2 * bLvl + bFt)}, pts.options().dtype(at::kByte));
        int64_t off = 0;
        auto piece = [&](int64_t bytes, int64_t used, at::ScalarType dt) {
            Tensor t = f->block.narrow(0, off, used).view(dt);
            off += bytes;
            return t;
        };
        f->mn = piece(bBox, B * 3 * 4, at::kFloat).view({B, 3});
        f->mx = piece(bBox, B * 3 * 4, at::kFloat).view({B, 3});
        f->sizes = piece(bSz, (L + 1) * 4, at::kInt);
        for (int l = 0; l < L; ++l) {
            f->ints.push_back(piece(bLvl, 3 * f->ca * 4, at::kInt));
            f->flts.push_back(piece(bLvl, 3 * f->ca * 4, at::kFloat));
            if (with_feats)
                f->lfeats.push_back(piece(bFt, f->ca * f->feats.size(1) * (int64_t)f->feats.element_size(), f->feats.scalar_type()));
        }
    }
    f->alloc_stream = (void*)ss;
    f->event = take_event();
    if (after_mode == 2 && after_event) {
        hip_check(hipStreamWaitEvent(ss, (hipEvent_t)(uintptr_t)after_event, 0), "hipStreamWaitEvent");
    } else if (after_mode != 1) {
        static thread_local hipEvent_t fork_ev = nullptr;
        if (!fork_ev) hip_check(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming), "hipEventCreate");
        hip_check(hipEventRecord(fork_ev, (hipStream_t)stream), "hipEventRecord");
        hip_check(hipStreamWaitEvent(ss, fork_ev, 0), "hipStreamWaitEvent");
    }
    if (Issuer::enabled()) {
        f->done.store(0, std::memory_order_release);
        Issuer::get(1).push([f, ss] { f->run(ss); });
    } else {
        f->run(ss);
    }
    return f;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, mod) {
    mod.doc() = "PyTorch-ROCm side of the native step executor of libmccnn_hip.so";
    if (mccnn::debug_int("trace_terminate", 0)) {   // debugging: where std::terminate was called from
        std::set_terminate([] {
            void* frames[64];
            const int n = backtrace(frames, 64);
            backtrace_symbols_fd(frames, n, 2);
            abort();
        });
    }
    static py::exception<CapacityError> cap_exc(mod, "CapacityError");
    py::register_exception_translator([](std::exception_ptr p) {
        try {
            if (p) std::rethrow_exception(p);
        } catch (const CapacityError& e) {
            PyErr_SetObject(cap_exc.ptr(), py::int_(e.edges).ptr());
        }
    });
    py::class_<Geo, std::shared_ptr<Geo>>(mod, "Geometry")
        .def_readonly("buf", &Geo::buf)
        .def_readonly("n", &Geo::n)
        .def_readonly("m", &Geo::m)
        .def_readonly("nc", &Geo::nc)
        .def_readonly("B", &Geo::B)
        .def_readonly("e_cap", &Geo::e_cap)
        .def_property_readonly("e", [](const Geo& g) { return g.e.load(std::memory_order_relaxed); })
        .def_readonly("grid_owner", &Geo::grid_owner)
        .def_readwrite("uses", &Geo::uses)
        .def_readonly("side", &Geo::side)
        .def_readonly("have", &Geo::have)
        // (none of the calls below touches a Python object, and all of them may wait -- for a helper thread's job, which can
        // in turn wait for a device-side edge total, or for the device itself: the GIL is released for their whole
        // duration, so data-loader threads keep running and a helper thread that drops the last reference to a tensor
        // with a Python wrapper can take the GIL in its destructor instead of dead-locking against a spinning caller)
        .def("join", [](Geo& g, const at::Tensor& like) { g.join(cur_stream(like)); }, py::call_guard<py::gil_scoped_release>())
        .def("prebuild", &Geo::prebuild, py::arg("what"), py::arg("avg"), py::arg("side"), py::arg("like"),
             py::call_guard<py::gil_scoped_release>())
        .def("edges", &Geo::edges, py::arg("wait_us") = -1, py::call_guard<py::gil_scoped_release>())
        .def("info", &Geo::info, py::call_guard<py::gil_scoped_release>());
    mod.def("build_geometry", &build_geometry, py::arg("pts"), py::arg("bids"), py::arg("centres"), py::arg("cbids"),
            py::arg("mn"), py::arg("mx"), py::arg("B"), py::arg("nc"), py::arg("radius"), py::arg("scale_inv"),
            py::arg("window"), py::arg("use_pdf"), py::arg("capacity"), py::arg("grid_from").none(true),
            py::arg("side") = -1, py::arg("fork") = false, py::arg("background") = false,
            py::arg("after").none(true) = py::none(), py::call_guard<py::gil_scoped_release>());
    mod.def("sampled_features", &sampled_features, py::call_guard<py::gil_scoped_release>());
    mod.def("begin_geometry_batch", &begin_geometry_batch);
    mod.def("end_geometry_batch", &end_geometry_batch, py::call_guard<py::gil_scoped_release>());
    mod.def("prebuild_async", &prebuild_async, py::arg("geometry"), py::arg("what"), py::arg("avg"),
            py::call_guard<py::gil_scoped_release>());
    mod.def("conv", &conv, py::arg("geometry"), py::arg("feats"), py::arg("w1"), py::arg("b1"), py::arg("w2"), py::arg("b2"),
            py::arg("w3"), py::arg("b3"), py::arg("fout"), py::arg("combin"), py::arg("avg"), py::arg("deterministic") = false,
            py::call_guard<py::gil_scoped_release>());
    mod.def("compute_aabb", &compute_aabb);
    mod.def("hierarchy_levels", &hierarchy_levels, py::call_guard<py::gil_scoped_release>());
    py::class_<HierFuture, std::shared_ptr<HierFuture>>(mod, "HierarchyFuture")
        .def("result", &HierFuture::result)
        .def("done", [](HierFuture& f) { return f.done.load(std::memory_order_acquire) != 0; });
    mod.def("hierarchy_prefetch", &hierarchy_prefetch, py::arg("pts"), py::arg("bids"), py::arg("radii"), py::arg("B"),
            py::arg("scale_inv"), py::arg("pmode"), py::arg("after_mode") = 0, py::arg("after_event") = 0,
            py::arg("feats") = py::none(), py::call_guard<py::gil_scoped_release>());
    mod.def("shutdown_helpers", [] { for (int k = 0; k < 3; ++k) Issuer::get(k).retire(); },
            py::call_guard<py::gil_scoped_release>());
    mod.def("wait_ns", [] { return (long long)g_wait_ns.load(std::memory_order_relaxed); });
    mod.def("debug_counters", [] {
        py::dict d;
        d["caller_orderings"] = (long long)g_caller_orderings.load(std::memory_order_relaxed);
        return d;
    });
    mod.def("debug_times", [](bool reset) {
        static const char* names[T_N] = {"fwd", "fwd_lib", "fwd_alloc", "bwd", "bwd_lib", "bwd_alloc", "bwd_views", "bwd_join"};
        py::dict d;
        for (int k = 0; k < T_N; ++k) {
            d[names[k]] = py::make_tuple((long long)g_t_ns[k].load(), (long long)g_t_calls[k].load());
            if (reset) { g_t_ns[k].store(0); g_t_calls[k].store(0); }
        }
        return d;
    }, py::arg("reset") = false);
}
