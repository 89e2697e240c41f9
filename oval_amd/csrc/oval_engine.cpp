// oval_engine.cpp — staging slots, micro-batcher and chunk stream (see
// oval_engine.hpp).
//
// A slot is everything one batch needs: a context (stream), a fill layout and
// a validate layout, a pinned host buffer and a device buffer per direction,
// pinned and device result arrays.  All of it is allocated once per slot and
// grows only; a batch allocates nothing.
//
// Micro-batching.  Callers reserve a job index and a byte range in the OPEN
// slot with one compare-and-swap on the slot's ticket word (no lock), then
// write their descriptor and (validate) their body straight into the slot's
// pinned memory.  A lane thread takes the open slot as soon as it holds work:
// it installs a fresh open slot, closes the ticket (which fixes the batch),
// waits for the callers still copying, and submits the batch on the slot's
// stream: a small batch is ONE kernel launch that reads and writes the pinned
// buffers over PCIe itself, a large one goes through the copy engines and the
// two streaming kernels.  While the lane waits for the device the next open
// slot fills up, so a batch is whatever arrived while the lanes were busy: with
// few callers that is one or two jobs (no added latency), with 255 it is
// dozens.  Every caller sleeps on a futex word of its own and is woken through
// a tree (see kWakeFanout); each copies its own body out of the pinned buffer,
// and the last one returns the slot to the free list.
// The pipeline lock is taken a few times per batch, never per job.
//
// Two things keep a call short when few are under way.  A call that finds no
// more calls inside execute() than the host has cores (at most 16) skips all of
// the above and launches a batch of one in a slot of its own: the hand-over to a
// lane costs two sleeps and two wake-ups, more than a launch.  And a small
// batch on a polling lane (or such a batch of one) ends without a stream event:
// the kernel's last workgroup stores the batch's sequence number to a pinned
// word once everything the batch wrote has reached host memory, and the host
// reads its own memory.
//
// Large work goes other ways: a batch form (run) beyond 16 MiB travels as
// sub-batches through up to three slots with its host copies on helper threads,
// and a single object of 8 MiB or more takes the sliced host-memory path of the
// C ABI on a slot's context (run_large).
#include "oval_engine.hpp"

#include <hip/hip_runtime.h>
#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "oval_capi_internal.h"
#include "oval_kernels.h"

namespace oval {
namespace engine {
namespace {

constexpr size_t kUnit = OVAL_UNIT_SIZE;
constexpr size_t kSharedBytes = 16u << 20;   // body capacity of a shared slot, per direction
constexpr size_t kSharedJobs = 2048;         // jobs of one kind a shared slot takes
constexpr size_t kSharedJobMax = 256u << 10;  // a bigger job runs in a slot of its own
// A batch of at most this many body bytes, made of jobs no larger than
// kSharedJobMax, runs as ONE kernel launch that reads and writes the pinned host
// buffers over PCIe itself (launch_host_batch); a larger one goes through the
// copy engines and the two streaming kernels, which sustain a higher rate.
constexpr size_t kZeroCopyMax = 4u << 20;
constexpr int kMaxLanes = 8;
constexpr int kDirectSlots = 16;             // calls that launch on their own hold one each (see direct_max)
constexpr int kSlots = 2 * kMaxLanes + 2 + kDirectSlots;   // in flight + being read out + open + one spare + direct

void futex_wait(std::atomic<uint32_t>* a, uint32_t seen)
{
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(a), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
}

void futex_wake(std::atomic<uint32_t>* a, int n)
{
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(a), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
}

// Waking the callers of a batch is a tree over its job numbers 0 .. n-1
// (validate jobs first, then fill jobs): whoever completes the batch wakes job
// 0, and the caller of job d, once awake, wakes jobs kWakeFanout d + 1 ..
// kWakeFanout d + kWakeFanout before it collects its outcome.  One wake-all
// from the lane thread costs it a few microseconds per sleeper (100+ us for a
// batch of 100 jobs), during which that lane submits nothing.
constexpr int kWakeFanout = 4, kWakeRoots = 1;
std::atomic<int> g_wake_fanout{kWakeFanout}, g_wake_roots{kWakeRoots};   // set_wake_tree(): benchmarks only
std::atomic<int> g_spin_ns{0};            // how long a caller polls its word before it sleeps on it
constexpr uint32_t kFlagMaxJobs = 24;
std::atomic<bool> g_flag_completion{true};   // small batches on polling lanes end by a flag in host memory, not by an event

uint64_t now_ns()
{
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

constexpr uint64_t kClosed = 1ull << 63;
constexpr uint64_t kUnitsMask = (1ull << 17) - 1, kJobsMask = (1ull << 12) - 1;
constexpr int kValUnitsShift = 0, kFillUnitsShift = 17, kValJobsShift = 34, kFillJobsShift = 46;
constexpr uint64_t kSharedUnits = kSharedBytes / kUnit;
static_assert(kSharedUnits <= kUnitsMask && kSharedJobs <= kJobsMask, "ticket fields too narrow");

inline uint32_t t_val_units(uint64_t t) { return (uint32_t)((t >> kValUnitsShift) & kUnitsMask); }
inline uint32_t t_fill_units(uint64_t t) { return (uint32_t)((t >> kFillUnitsShift) & kUnitsMask); }
inline uint32_t t_val_jobs(uint64_t t) { return (uint32_t)((t >> kValJobsShift) & kJobsMask); }
inline uint32_t t_fill_jobs(uint64_t t) { return (uint32_t)((t >> kFillJobsShift) & kJobsMask); }
inline uint32_t t_jobs(uint64_t t) { return t_val_jobs(t) + t_fill_jobs(t); }

struct alignas(64) WakeWord {
    std::atomic<uint32_t> v{0};
};

struct Slot {
    oval_ctx* ctx = nullptr;
    oval_layout* fill_layout = nullptr;
    oval_layout* val_layout = nullptr;
    hipEvent_t done_spin = nullptr, done_block = nullptr;
    // A small batch on a polling lane ends without an event: the kernel's last
    // workgroup stores the batch's sequence number to done_flag (pinned, a cache
    // line of its own) once everything it wrote has reached host memory.
    uint32_t* done_flag = nullptr;
    uint32_t* done_count_dev = nullptr;   // workgroups finished; zero between launches
    uint32_t done_seq = 0;
    bool by_flag = false;                 // the batch in flight signals through done_flag
    hipEvent_t half_ev[2] = {nullptr, nullptr};               // chunk stream: one per half
    uint8_t* up_host = nullptr;   void* up_dev = nullptr;   size_t up_cap = 0;     // validate bodies
    uint8_t* down_host = nullptr; void* down_dev = nullptr; size_t down_cap = 0;   // fill bodies
    oval_obj_result* res_host = nullptr; oval_obj_result* res_dev = nullptr;
    size_t job_cap = 0;
    oval_obj_desc* fdescs = nullptr;      // pinned, job_cap each
    oval_obj_desc* vdescs = nullptr;
    // the batch being assembled or in flight
    uint32_t n_fill = 0, n_val = 0;
    size_t fill_bytes = 0, val_bytes = 0;
    size_t max_job_bytes = 0;
    // Reservations of the open slot: one word, advanced by compare-and-swap.
    // bits 0-16 validate units, 17-33 fill units, 34-45 validate jobs, 46-57
    // fill jobs, bit 63 closed (set whenever the slot is not the open one).
    std::atomic<uint64_t> ticket{kClosed};
    std::atomic<uint32_t> written{0};     // callers that have finished writing their descriptor / body
    std::atomic<int> readers{0};          // callers that have not collected their outcome yet
    // One futex word per job (validate jobs first, then fill jobs), each in a
    // cache line of its own: a caller sleeps on its own word, so the sleepers of
    // a batch do not queue on one kernel futex bucket, and is woken by its parent
    // in a tree over the batch's job numbers (see wake_children).
    std::unique_ptr<WakeWord[]> words;
    int wake_fanout = kWakeFanout, wake_roots = kWakeRoots;   // shape of the batch's wake tree, fixed when it closes
    bool failed = false;
    std::string err;
};

struct Pipeline {
    std::mutex mu;                        // open slot, free list, batching state, statistics, err
    std::condition_variable lane_cv;      // lanes: an open slot holds work (or stop)
    std::condition_variable slot_cv;      // callers: a slot was freed or the open one was taken
    Slot slots[kSlots];
    std::vector<Slot*> free_slots;
    std::atomic<Slot*> open{nullptr};     // written under mu, read without
    int slot_waiters = 0;                 // callers asleep on slot_cv
    int device = 0;
    bool inited = false;
    std::atomic<bool> enabled{false};
    bool stop = false;
    int window_us = 0;
    int n_lanes = 2, blocking = 0;
    std::vector<std::thread> lanes;
    uint64_t batches = 0, jobs = 0;
    uint64_t prof[kProfileWords] = {};    // lane time by phase, ns (lane_profile)
    std::string err;
    std::mutex ctl_mu;                    // serialises set_batching / shutdown (held across the joins)
};

// Never destroyed: a process that exits while micro-batching is on still has lane threads asleep in
// lane_loop, and destroying their std::thread objects (or the mutex they wait on) at exit would abort it.
Pipeline& g = *new Pipeline;

void pool_drop_all();                     // pinned buffers of the generated-parts pool (below)

int slot_fail(Slot& s, int rc)
{
    s.err = oval_last_error();
    s.failed = true;
    return rc;
}

int hip_fail(Slot& s, hipError_t e, const char* what)
{
    s.err = std::string(what) + ": " + hipGetErrorString(e);
    s.failed = true;
    return OVAL_E_HIP;
}

size_t round_cap(size_t bytes)
{
    size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
    return (cap + 65535) & ~(size_t)65535;
}

// Grow-only; called by the slot's current owner with no work of the slot in flight.
int ensure_slot(Slot& s, size_t up, size_t down, size_t jobs)
{
    int rc;
    if (!s.ctx) {
        if ((rc = oval_ctx_create(g.device, nullptr, &s.ctx))) return slot_fail(s, rc);
        if ((rc = oval_layout_create(s.ctx, nullptr, 0, &s.fill_layout))) return slot_fail(s, rc);
        if ((rc = oval_layout_create(s.ctx, nullptr, 0, &s.val_layout))) return slot_fail(s, rc);
        hipError_t e = hipEventCreateWithFlags(&s.done_spin, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done_block, hipEventDisableTiming | hipEventBlockingSync);
        for (int h = 0; h < 2 && e == hipSuccess; ++h) e = hipEventCreateWithFlags(&s.half_ev[h], hipEventDisableTiming);
        if (e != hipSuccess) return hip_fail(s, e, "hipEventCreate");
        if ((rc = oval_internal_pinned_alloc_coherent(s.ctx, 64, (void**)&s.done_flag))) return slot_fail(s, rc);
        if ((rc = oval_dev_alloc(s.ctx, 64, (void**)&s.done_count_dev))) return slot_fail(s, rc);
        *s.done_flag = 0;
        s.done_seq = 0;
        if ((e = hipMemset(s.done_count_dev, 0, 64)) != hipSuccess) return hip_fail(s, e, "hipMemset");
    }
    if (up > s.up_cap) {
        if (s.up_host) oval_pinned_free(s.ctx, s.up_host);
        if (s.up_dev) oval_dev_free(s.ctx, s.up_dev);
        s.up_host = nullptr; s.up_dev = nullptr; s.up_cap = 0;
        const size_t cap = round_cap(up);
        if ((rc = oval_internal_pinned_alloc_coherent(s.ctx, cap, (void**)&s.up_host))) return slot_fail(s, rc);
        if ((rc = oval_dev_alloc(s.ctx, cap, &s.up_dev))) return slot_fail(s, rc);
        s.up_cap = cap;
    }
    if (down > s.down_cap) {
        if (s.down_host) oval_pinned_free(s.ctx, s.down_host);
        if (s.down_dev) oval_dev_free(s.ctx, s.down_dev);
        s.down_host = nullptr; s.down_dev = nullptr; s.down_cap = 0;
        const size_t cap = round_cap(down);
        if ((rc = oval_internal_pinned_alloc_coherent(s.ctx, cap, (void**)&s.down_host))) return slot_fail(s, rc);
        if ((rc = oval_dev_alloc(s.ctx, cap, &s.down_dev))) return slot_fail(s, rc);
        s.down_cap = cap;
    }
    if (jobs > s.job_cap) {
        if (s.res_host) oval_pinned_free(s.ctx, s.res_host);
        if (s.res_dev) oval_dev_free(s.ctx, s.res_dev);
        s.res_host = nullptr; s.res_dev = nullptr; s.job_cap = 0;
        const size_t cap = jobs < 64 ? 64 : jobs + jobs / 4;
        if ((rc = oval_internal_pinned_alloc_coherent(s.ctx, cap * sizeof(oval_obj_result), (void**)&s.res_host))) return slot_fail(s, rc);
        if ((rc = oval_dev_alloc(s.ctx, cap * sizeof(oval_obj_result), (void**)&s.res_dev))) return slot_fail(s, rc);
        if (s.fdescs) oval_pinned_free(s.ctx, s.fdescs);
        s.fdescs = nullptr; s.vdescs = nullptr;
        if ((rc = oval_internal_pinned_alloc_coherent(s.ctx, 2 * cap * sizeof(oval_obj_desc), (void**)&s.fdescs))) return slot_fail(s, rc);
        s.vdescs = s.fdescs + cap;
        s.job_cap = cap;
        if (jobs >= kSharedJobs && !s.words) s.words.reset(new WakeWord[2 * kSharedJobs]);
    }
    return OVAL_OK;
}

void destroy_slot(Slot& s)
{
    if (!s.ctx) return;
    oval_ctx_sync(s.ctx);
    if (s.up_host) oval_pinned_free(s.ctx, s.up_host);
    if (s.up_dev) oval_dev_free(s.ctx, s.up_dev);
    if (s.down_host) oval_pinned_free(s.ctx, s.down_host);
    if (s.down_dev) oval_dev_free(s.ctx, s.down_dev);
    if (s.res_host) oval_pinned_free(s.ctx, s.res_host);
    if (s.res_dev) oval_dev_free(s.ctx, s.res_dev);
    if (s.fdescs) oval_pinned_free(s.ctx, s.fdescs);
    if (s.done_flag) oval_pinned_free(s.ctx, s.done_flag);
    if (s.done_count_dev) oval_dev_free(s.ctx, s.done_count_dev);
    s.done_flag = nullptr; s.done_count_dev = nullptr;
    if (s.fill_layout) oval_layout_destroy(s.ctx, s.fill_layout);
    if (s.val_layout) oval_layout_destroy(s.ctx, s.val_layout);
    if (s.done_spin) (void)hipEventDestroy(s.done_spin);
    if (s.done_block) (void)hipEventDestroy(s.done_block);
    for (hipEvent_t& e : s.half_ev)
        if (e) (void)hipEventDestroy(e);
    oval_ctx_destroy(s.ctx);
    s.ctx = nullptr; s.fill_layout = nullptr; s.val_layout = nullptr;
    s.done_spin = nullptr; s.done_block = nullptr; s.half_ev[0] = nullptr; s.half_ev[1] = nullptr;
    s.up_host = nullptr; s.up_dev = nullptr; s.up_cap = 0;
    s.down_host = nullptr; s.down_dev = nullptr; s.down_cap = 0;
    s.res_host = nullptr; s.res_dev = nullptr; s.job_cap = 0;
    s.fdescs = nullptr; s.vdescs = nullptr;
}

void reset_batch(Slot& s)
{
    s.n_fill = 0; s.n_val = 0; s.fill_bytes = 0; s.val_bytes = 0; s.max_job_bytes = 0;
    s.written.store(0, std::memory_order_relaxed);
    s.failed = false;
    s.err.clear();
}

// Everything a batch needs, enqueued on the slot's stream; no host wait.
int submit(Slot& s, bool blocking)
{
    int rc;
    hipStream_t stream = oval_internal_ctx_stream(s.ctx);
    s.by_flag = false;
    if (s.max_job_bytes <= kSharedJobMax && s.val_bytes + s.fill_bytes <= kZeroCopyMax) {
        // (every workgroup pays a system-wide release of its stores for the flag: with a few dozen of them
        // that costs the device more than the event saves the host — 38 jobs: 22.5 against 19.0 us)
        const uint32_t n_jobs = s.n_val + s.n_fill;
        const bool flag = !blocking && g_flag_completion.load(std::memory_order_relaxed) && n_jobs != 0 && n_jobs <= kFlagMaxJobs;
        if (flag) ++s.done_seq;
        const hipError_t e = launch_host_batch(stream, s.vdescs, s.n_val, s.fdescs, s.n_fill, s.up_host, s.down_host,
                                               s.res_host, flag ? s.done_count_dev : nullptr,
                                               flag ? s.done_flag : nullptr, s.done_seq);
        if (e != hipSuccess) return hip_fail(s, e, "host batch launch");
        if (flag) {
            s.by_flag = true;
            return OVAL_OK;                                      // no event: wait_done reads the flag
        }
    } else {
    if (s.n_val) {
        if ((rc = oval_copy_h2d(s.ctx, s.up_dev, s.up_host, s.val_bytes))) return slot_fail(s, rc);
        if ((rc = oval_layout_reset(s.ctx, s.val_layout, s.vdescs, s.n_val))) return slot_fail(s, rc);
        if ((rc = oval_validate(s.ctx, s.val_layout, s.up_dev, s.up_cap, s.res_dev, nullptr))) return slot_fail(s, rc);
        if ((rc = oval_copy_d2h(s.ctx, s.res_host, s.res_dev, s.n_val * sizeof(oval_obj_result)))) return slot_fail(s, rc);
    }
    if (s.n_fill) {
        if ((rc = oval_layout_reset(s.ctx, s.fill_layout, s.fdescs, s.n_fill))) return slot_fail(s, rc);
        if ((rc = oval_fill(s.ctx, s.fill_layout, s.down_dev, s.down_cap))) return slot_fail(s, rc);
        if ((rc = oval_copy_d2h(s.ctx, s.down_host, s.down_dev, s.fill_bytes))) return slot_fail(s, rc);
    }
    }
    const hipError_t e = hipEventRecord(blocking ? s.done_block : s.done_spin, stream);
    if (e != hipSuccess) return hip_fail(s, e, "hipEventRecord");
    return OVAL_OK;
}

int wait_done(Slot& s, bool blocking)
{
    if (s.by_flag) {
        s.by_flag = false;
        volatile uint32_t* flag = s.done_flag;
        for (uint64_t spins = 0; *flag != s.done_seq; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 0xFFFFu) == 0xFFFFu) {
                // every few milliseconds: a stream that has gone idle or bad without the flag would spin here for ever
                const hipError_t q = hipStreamQuery(oval_internal_ctx_stream(s.ctx));
                if (q == hipSuccess) {
                    if (*flag == s.done_seq) break;
                    return hip_fail(s, hipErrorUnknown, "the batch ended without its completion flag");
                }
                if (q != hipErrorNotReady) return hip_fail(s, q, "hipStreamQuery");
                (void)hipGetLastError();
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        // the runtime retires finished launches when it is asked about the stream; nothing else asks on this path
        if ((s.done_seq & 0xFFu) == 0u) {
            (void)hipStreamQuery(oval_internal_ctx_stream(s.ctx));
            (void)hipGetLastError();
        }
        return OVAL_OK;
    }
    const hipError_t e = hipEventSynchronize(blocking ? s.done_block : s.done_spin);
    if (e != hipSuccess) return hip_fail(s, e, "hipEventSynchronize");
    return OVAL_OK;
}

void note_failure(const Slot& s)
{
    std::lock_guard<std::mutex> lk(g.mu);
    g.err = s.err;
}

void ensure_inited_locked()
{
    if (g.inited) return;
    for (Slot& s : g.slots) g.free_slots.push_back(&s);
    g.inited = true;
}

// Generated bodies (parts_open) keep their slot until their last slice has landed, but nobody is told
// when that happens: whoever finds no slot free looks whether one of them is done (defined below).
bool reap_finished_parts();

// A free slot for the caller's exclusive use.
Slot* acquire_slot()
{
    std::unique_lock<std::mutex> lk(g.mu);
    ensure_inited_locked();
    while (g.free_slots.empty()) {
        lk.unlock();
        const bool reaped = reap_finished_parts();
        lk.lock();
        if (reaped || !g.free_slots.empty()) continue;
        g.slot_waiters++;
        g.slot_cv.wait_for(lk, std::chrono::microseconds(200));   // a landed body frees its slot without a notify
        g.slot_waiters--;
    }
    Slot* s = g.free_slots.back();
    g.free_slots.pop_back();
    return s;
}

// A free slot if there is one right now.
Slot* try_acquire_slot()
{
    std::lock_guard<std::mutex> lk(g.mu);
    ensure_inited_locked();
    if (g.free_slots.empty()) return nullptr;
    Slot* s = g.free_slots.back();
    g.free_slots.pop_back();
    return s;
}

// Makes `c` the open slot (mu held).  Its first shared use allocates its buffers.
bool install_open_locked(Slot* c)
{
    if (ensure_slot(*c, kSharedBytes, kSharedBytes, kSharedJobs)) {
        g.err = c->err;
        return false;
    }
    reset_batch(*c);
    c->ticket.store(0, std::memory_order_relaxed);
    g.open.store(c, std::memory_order_release);
    return true;
}

void release_slot(Slot* s)
{
    bool wake;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        // callers may be waiting for an open slot because every slot was busy
        if (!(g.enabled.load(std::memory_order_relaxed) && !g.stop && !g.open.load(std::memory_order_relaxed) &&
              install_open_locked(s)))
            g.free_slots.push_back(s);
        wake = g.slot_waiters > 0;
    }
    if (wake) g.slot_cv.notify_all();
}

void wake_job(Slot* s, uint32_t d)
{
    std::atomic<uint32_t>* w = &s->words[d < s->n_val ? d : kSharedJobs + (d - s->n_val)].v;
    w->fetch_add(1, std::memory_order_release);
    futex_wake(w, 1);
}

// Job d wakes jobs roots + fanout d ... roots + fanout d + fanout - 1; jobs 0 ... roots - 1 are woken
// by whoever completed the batch.
void wake_children(Slot* s, uint32_t d)
{
    const uint32_t n = s->n_val + s->n_fill;
    const uint32_t f = (uint32_t)s->wake_fanout, r = (uint32_t)s->wake_roots;
    for (uint32_t c = r + f * d, k = 0; k < f && c < n; ++c, ++k) wake_job(s, c);
}

// Everything is read from the slot before the first wake: once the last root has
// been woken the batch's callers may all be through, and the slot released,
// reopened and described anew, before this function returns.
void wake_roots(Slot* s)
{
    const uint32_t n_val = s->n_val, n = s->n_val + s->n_fill;
    uint32_t roots = (uint32_t)s->wake_roots;
    if (roots > n) roots = n;
    WakeWord* const words = s->words.get();
    for (uint32_t c = 0; c < roots; ++c) {
        std::atomic<uint32_t>* w = &words[c < n_val ? c : kSharedJobs + (c - n_val)].v;
        w->fetch_add(1, std::memory_order_release);
        futex_wake(w, 1);
    }
}

// The batch of a slot whose ticket was closed at `t`: waits for the callers
// still copying in, runs it, publishes the outcome and wakes the callers.
void process_batch(Slot* s, uint64_t t, bool blocking, uint64_t* phase_ns)
{
    s->n_val = t_val_jobs(t);
    s->n_fill = t_fill_jobs(t);
    s->val_bytes = (size_t)t_val_units(t) * kUnit;
    s->fill_bytes = (size_t)t_fill_units(t) * kUnit;
    s->max_job_bytes = 0;                                        // every job of a shared slot is small
    s->wake_fanout = g_wake_fanout.load(std::memory_order_relaxed);
    s->wake_roots = g_wake_roots.load(std::memory_order_relaxed);
    const uint32_t n_jobs = s->n_val + s->n_fill;
    const uint64_t t0 = now_ns();
    for (int spin = 0; s->written.load(std::memory_order_acquire) != n_jobs; ++spin)
        if (spin > 64) sched_yield();
    const uint64_t t1 = now_ns();
    int rc = submit(*s, blocking);
    const uint64_t t2 = now_ns();
    if (!rc) rc = wait_done(*s, blocking);
    else (void)oval_ctx_sync(s->ctx);                            // whatever was enqueued before the failure
    const uint64_t t3 = now_ns();
    if (rc) note_failure(*s);
    s->readers.store((int)n_jobs, std::memory_order_relaxed);
    wake_roots(s);
    if (phase_ns) {
        phase_ns[0] = t1 - t0; phase_ns[1] = t2 - t1; phase_ns[2] = t3 - t2; phase_ns[3] = now_ns() - t3;
    }
}

void lane_loop()
{
    std::unique_lock<std::mutex> lk(g.mu);
    (void)hipSetDevice(g.device);                                // a new thread starts on device 0
    for (;;) {
        g.lane_cv.wait(lk, [] {
            Slot* o = g.open.load(std::memory_order_relaxed);
            return g.stop || (o && t_jobs(o->ticket.load(std::memory_order_acquire)) != 0);
        });
        if (g.stop) return;                                      // set_batching(false) drains what is left
        Slot* s = g.open.load(std::memory_order_relaxed);
        // A window > 0 holds the door open that long for callers arriving together.
        if (g.window_us > 0) {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(g.window_us);
            while (!g.stop && g.open.load(std::memory_order_relaxed) == s &&
                   g.lane_cv.wait_until(lk, deadline) != std::cv_status::timeout) {
            }
            if (g.stop) return;
            if (g.open.load(std::memory_order_relaxed) != s) continue;   // another lane took it meanwhile
        }
        // the successor first, so that callers always find an open slot; then close this one
        g.open.store(nullptr, std::memory_order_release);
        if (!g.free_slots.empty()) {
            Slot* f = g.free_slots.back();
            g.free_slots.pop_back();
            if (!install_open_locked(f)) g.free_slots.push_back(f);
        }
        const uint64_t t = s->ticket.fetch_or(kClosed, std::memory_order_acq_rel);
        const bool blocking = g.blocking != 0;
        const bool wake = g.slot_waiters > 0;
        lk.unlock();
        if (wake) g.slot_cv.notify_all();                        // callers that found the slot full
        uint64_t ph[4];
        process_batch(s, t, blocking, ph);
        const uint64_t t4 = now_ns();
        lk.lock();
        g.batches++;
        g.jobs += t_jobs(t);
        for (int i = 0; i < 4; ++i) g.prof[i] += ph[i];
        g.prof[4] += now_ns() - t4;                              // re-taking the pipeline lock
        g.prof[5] += 1;
        g.prof[6] += ((uint64_t)t_val_units(t) + t_fill_units(t)) * kUnit;
    }
}

}  // namespace

int init(int device)
{
    {
        std::lock_guard<std::mutex> lk(g.mu);
        ensure_inited_locked();
        bool any = false;
        for (const Slot& s : g.slots) any |= s.ctx != nullptr;
        if (!any) g.device = device;                             // the first call decides
    }
    // one slot's context proves the device is usable
    Slot* s = acquire_slot();
    const int rc = ensure_slot(*s, 0, 0, 0);
    if (rc) note_failure(*s);
    release_slot(s);
    return rc;
}

std::string last_error()
{
    std::lock_guard<std::mutex> lk(g.mu);
    return g.err;
}

namespace {

// A batch larger than this is cut into sub-batches that travel through several
// slots at once: while one is on the device the next one's bodies are gathered
// into pinned memory and the one before is scattered back to the callers'
// buffers, so the host copies, the PCIe copies and the kernels overlap and the
// pinned memory a batch needs stays bounded whatever its size.
constexpr size_t kRunChunkBytes = 16u << 20;   // 4 ... 64 MiB measured alike within noise; below 4 the per-sub-batch calls show
constexpr size_t kRunChunkJobs = 16384;
constexpr size_t kRunDepthMax = 6, kRunDepth = 3;
std::atomic<size_t> g_run_chunk_bytes{kRunChunkBytes}, g_run_depth{kRunDepth};   // set_run_shape(): benchmarks only
// host copies of a sub-batch are spread over the helper threads from this size on
constexpr size_t kParallelCopyBytes = 2u << 20;

// fn(first, last) over [0, n) in up to eight ranges of similar byte counts
template <class Fn>
void for_job_ranges(Job* const* jobs, size_t n, size_t bytes, const Fn& fn)
{
    if (bytes < kParallelCopyBytes || n < 2) {
        fn((size_t)0, n);
        return;
    }
    const size_t parts = n < 8 ? n : 8;
    std::vector<size_t> cut(parts + 1, n);
    cut[0] = 0;
    size_t acc = 0, k = 1;
    for (size_t i = 0; i < n && k < parts; ++i) {
        acc += (size_t)jobs[i]->desc.n_units * kUnit;
        if (acc * parts >= bytes * k) cut[k++] = i + 1;
    }
    oval_internal_parallel_for(parts, [&](size_t p) {
        if (cut[p] < cut[p + 1]) fn(cut[p], cut[p + 1]);
    });
}

struct SubBatch {
    Slot* s = nullptr;
    Job* const* jobs = nullptr;
    size_t n = 0;
    int rc = OVAL_OK;
};

// Lays the jobs out in the slot, gathers the bodies to validate and enqueues everything; no wait.
void sub_start(SubBatch& b)
{
    Slot* s = b.s;
    reset_batch(*s);
    size_t up = 0, down = 0, nf = 0, nv = 0;
    for (size_t i = 0; i < b.n; ++i) {
        const size_t bytes = (size_t)b.jobs[i]->desc.n_units * kUnit;
        if (b.jobs[i]->is_fill) { down += bytes; nf++; } else { up += bytes; nv++; }
        if (bytes > s->max_job_bytes) s->max_job_bytes = bytes;
    }
    b.rc = ensure_slot(*s, up, down, nf > nv ? nf : nv);
    if (b.rc) return;
    for (size_t i = 0; i < b.n; ++i) {
        Job* j = b.jobs[i];
        const size_t bytes = (size_t)j->desc.n_units * kUnit;
        if (j->is_fill) {
            j->desc.body_offset = s->fill_bytes;
            s->fdescs[s->n_fill++] = j->desc;
            s->fill_bytes += bytes;
        } else {
            j->desc.body_offset = s->val_bytes;
            s->vdescs[s->n_val++] = j->desc;
            s->val_bytes += bytes;
        }
    }
    if (s->n_val)
        for_job_ranges(b.jobs, b.n, up + down, [&](size_t first, size_t last) {
            for (size_t i = first; i < last; ++i) {
                const Job* j = b.jobs[i];
                if (!j->is_fill) std::memcpy(s->up_host + j->desc.body_offset, j->in, (size_t)j->desc.n_units * kUnit);
            }
        });
    b.rc = submit(*s, false);
    if (b.rc) (void)oval_ctx_sync(s->ctx);                       // whatever was enqueued before the failure
}

// Waits for the sub-batch, hands bodies and verdicts to the jobs and gives the slot back.
void sub_finish(SubBatch& b)
{
    Slot* s = b.s;
    if (!b.rc) b.rc = wait_done(*s, false);
    if (b.rc) {
        note_failure(*s);
        for (size_t i = 0; i < b.n; ++i) { b.jobs[i]->failed = true; b.jobs[i]->err = s->err; }
    } else {
        uint32_t vi = 0;
        for (size_t i = 0; i < b.n; ++i)
            if (!b.jobs[i]->is_fill) b.jobs[i]->res = s->res_host[vi++];
        if (s->n_fill)
            for_job_ranges(b.jobs, b.n, s->fill_bytes + s->val_bytes, [&](size_t first, size_t last) {
                for (size_t i = first; i < last; ++i) {
                    Job* j = b.jobs[i];
                    if (j->is_fill) std::memcpy(j->out, s->down_host + j->desc.body_offset, (size_t)j->desc.n_units * kUnit);
                }
            });
    }
    release_slot(s);
    b.s = nullptr;
}

}  // namespace

void run(Job* const* jobs, size_t n)
{
    if (n == 0) return;
    SubBatch ring[kRunDepthMax];
    const size_t depth = g_run_depth.load(std::memory_order_relaxed);
    const size_t chunk_bytes = g_run_chunk_bytes.load(std::memory_order_relaxed);
    size_t head = 0, live = 0;                                   // ring[head .. head + live) are on the device
    for (size_t first = 0; first < n;) {
        // the next sub-batch: jobs [first, last); a job larger than a sub-batch travels alone
        size_t last = first, bytes = 0;
        while (last < n && last - first < kRunChunkJobs) {
            const size_t b = (size_t)jobs[last]->desc.n_units * kUnit;
            if (last > first && bytes + b > chunk_bytes) break;
            bytes += b;
            ++last;
        }
        // A caller never waits for a slot while it holds one (many callers doing so could hold all
        // of them between them): if none is free it completes its own oldest sub-batch first.
        Slot* slot = live < depth ? try_acquire_slot() : nullptr;
        while (!slot && live) {
            sub_finish(ring[head]);
            head = (head + 1) % depth;
            --live;
            slot = try_acquire_slot();
        }
        if (!slot) slot = acquire_slot();
        SubBatch& b = ring[(head + live) % depth];
        b.s = slot;
        b.jobs = jobs + first;
        b.n = last - first;
        b.rc = OVAL_OK;
        sub_start(b);
        ++live;
        first = last;
    }
    for (; live; --live, head = (head + 1) % depth) sub_finish(ring[head]);
}

namespace {
// Calls inside execute() right now.  While there are no more of them than host
// cores to spin on, a launch of one's own is quicker than the hand-over to a lane
// and back (two sleeps and two wake-ups per call); beyond that the shared launch
// wins.  Put-cycles/s on a 16-core host, own launch against micro-batcher: 1
// thread 46 K / 23 K, 4: 157 K / 76 K, 8: 260 K / 159 K, 16: 310 K / 262 K, 32: 257 K /
// 430 K.
std::atomic<int> g_callers{0};
std::atomic<int> g_direct_max{-1};        // -1: min(16, host cores), settled at first use
std::atomic<uint64_t> g_direct_jobs{0};   // calls that launched on their own while micro-batching was on
int direct_max()
{
    int m = g_direct_max.load(std::memory_order_relaxed);
    if (m < 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        m = hw == 0 || hw > 16 ? 16 : (int)hw;
        g_direct_max.store(m, std::memory_order_relaxed);
    }
    return m;
}
struct CallerCount {
    int now;
    CallerCount() : now(g_callers.fetch_add(1, std::memory_order_relaxed) + 1) {}
    ~CallerCount() { g_callers.fetch_sub(1, std::memory_order_relaxed); }
};
}  // namespace

void set_direct_max(int callers) { g_direct_max.store(callers < 0 ? 0 : callers, std::memory_order_relaxed); }

namespace {

// One object of many megabytes in the caller's own (usually pageable) memory: the
// host-memory entry points move it in 32 MiB slices through the slot context's
// staging ring, pageable memory through pinned bounce slices filled on several
// threads, with copies and kernel overlapped.  A slot of its own sized for the
// whole body would pin that much memory and copy it one step after the other.
constexpr size_t kLargeJobBytes = 8u << 20;

void run_large(Job& job)
{
    Slot* s = acquire_slot();
    reset_batch(*s);
    int rc = ensure_slot(*s, 0, 0, 0);
    if (!rc) {
        oval_obj_desc d = job.desc;
        d.body_offset = 0;
        const uint64_t bytes = (uint64_t)d.n_units * kUnit;
        if (job.is_fill) rc = oval_fill_batch_host(s->ctx, &d, 1, job.out, bytes);
        else rc = oval_validate_batch_host(s->ctx, &d, 1, job.in, bytes, &job.res, nullptr);
        if (rc) slot_fail(*s, rc);
    }
    if (rc) {
        note_failure(*s);
        job.failed = true;
        job.err = s->err;
        if (s->ctx) (void)oval_ctx_sync(s->ctx);
    }
    release_slot(s);
}

}  // namespace

void execute(Job& job)
{
    if ((size_t)job.desc.n_units * kUnit >= kLargeJobBytes) {
        run_large(job);
        return;
    }
    const CallerCount here;
    if (here.now <= direct_max()) {
        Job* p = &job;
        run(&p, 1);
        if (g.enabled.load(std::memory_order_relaxed)) g_direct_jobs.fetch_add(1, std::memory_order_relaxed);
        return;
    }
    const uint64_t units = job.desc.n_units;
    const size_t bytes = (size_t)units * kUnit;
    const int ushift = job.is_fill ? kFillUnitsShift : kValUnitsShift;
    const int jshift = job.is_fill ? kFillJobsShift : kValJobsShift;
    Slot* s = nullptr;
    uint64_t before = 0;
    while (!s) {
        if (!g.enabled.load(std::memory_order_acquire) || bytes > kSharedJobMax) break;
        Slot* c = g.open.load(std::memory_order_acquire);
        bool full = false;
        if (c) {
            uint64_t t = c->ticket.load(std::memory_order_acquire);
            while (!(t & kClosed)) {
                if (((t >> ushift) & kUnitsMask) + units > kSharedUnits || ((t >> jshift) & kJobsMask) + 1 > kSharedJobs) {
                    full = true;
                    break;
                }
                if (c->ticket.compare_exchange_weak(t, t + (units << ushift) + (1ull << jshift),
                                                    std::memory_order_acq_rel, std::memory_order_acquire)) {
                    s = c;
                    before = t;
                    break;
                }
            }
            if (s || (!full && g.open.load(std::memory_order_acquire) != c)) continue;   // reserved, or a newer open slot
        }
        // no open slot (all busy), or it is full: install one, or sleep until a lane or a reader changes that
        std::unique_lock<std::mutex> lk(g.mu);
        if (!g.enabled.load(std::memory_order_relaxed)) break;
        Slot* o = g.open.load(std::memory_order_relaxed);
        if (!o) {
            if (!g.free_slots.empty()) {
                Slot* f = g.free_slots.back();
                g.free_slots.pop_back();
                if (!install_open_locked(f)) {
                    job.failed = true;
                    job.err = f->err;
                    g.free_slots.push_back(f);
                    return;
                }
                continue;
            }
        } else if (o != c || !full) {
            continue;                                            // it changed while the lock was taken
        }
        lk.unlock();
        if (reap_finished_parts()) continue;                     // a landed body was still holding a slot
        lk.lock();
        if (g.open.load(std::memory_order_relaxed) != o) continue;
        g.slot_waiters++;
        g.slot_cv.wait_for(lk, std::chrono::microseconds(200));
        g.slot_waiters--;
    }
    if (!s) {
        Job* p = &job;
        run(&p, 1);
        return;
    }
    // The batch cannot complete before this caller has written, so the value of
    // its futex word read here is the one its waker will change.
    const uint32_t idx = (uint32_t)((before >> jshift) & kJobsMask);
    std::atomic<uint32_t>* word = &s->words[job.is_fill ? kSharedJobs + idx : idx].v;
    const uint32_t seq = word->load(std::memory_order_acquire);
    const size_t off = (size_t)((before >> ushift) & kUnitsMask) * kUnit;
    job.desc.body_offset = off;
    (job.is_fill ? s->fdescs : s->vdescs)[idx] = job.desc;
    if (!job.is_fill) std::memcpy(s->up_host + off, job.in, bytes);
    s->written.fetch_add(1, std::memory_order_release);
    if (t_jobs(before) == 0) {
        // first job of the batch: a lane may be asleep (taking the lock orders
        // this against a lane that has just found the slot empty).  Done after the
        // job is written: between reservation and that point a caller holds up
        // its whole batch, so nothing that can sleep or be preempted belongs there.
        { std::lock_guard<std::mutex> lk(g.mu); }
        g.lane_cv.notify_one();
    }
    if (const int spin_ns = g_spin_ns.load(std::memory_order_relaxed)) {
        const uint64_t until = now_ns() + (uint64_t)spin_ns;
        while (word->load(std::memory_order_acquire) == seq && now_ns() < until) __builtin_ia32_pause();
    }
    while (word->load(std::memory_order_acquire) == seq) futex_wait(word, seq);
    wake_children(s, job.is_fill ? s->n_val + idx : idx);        // pass the wake-up on
    if (s->failed) {
        job.failed = true;
        job.err = s->err;
    } else if (job.is_fill) {
        std::memcpy(job.out, s->down_host + off, bytes);
    } else {
        job.res = s->res_host[idx];
    }
    if (s->readers.fetch_sub(1, std::memory_order_acq_rel) == 1) release_slot(s);
}

void set_lanes(int lanes, int blocking)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (lanes >= 1) g.n_lanes = lanes > kMaxLanes ? kMaxLanes : lanes;
    if (blocking >= 0) g.blocking = blocking ? 1 : 0;
}

void set_run_shape(size_t chunk_bytes, int depth)
{
    if (chunk_bytes >= (64u << 10)) g_run_chunk_bytes.store(chunk_bytes, std::memory_order_relaxed);
    if (depth >= 1) g_run_depth.store((size_t)depth > kRunDepthMax ? kRunDepthMax : (size_t)depth, std::memory_order_relaxed);
}

void set_flag_completion(bool on) { g_flag_completion.store(on, std::memory_order_relaxed); }

void set_caller_spin(int ns) { g_spin_ns.store(ns < 0 ? 0 : ns, std::memory_order_relaxed); }

void set_wake_tree(int fanout, int roots)
{
    if (fanout >= 1) g_wake_fanout.store(fanout > 64 ? 64 : fanout, std::memory_order_relaxed);
    if (roots >= 1) g_wake_roots.store(roots > 64 ? 64 : roots, std::memory_order_relaxed);
}

void set_batching(bool enable, int window_us, size_t /*max_bytes*/)
{
    std::lock_guard<std::mutex> ctl(g.ctl_mu);
    std::unique_lock<std::mutex> lk(g.mu);
    ensure_inited_locked();
    if (window_us >= 0) g.window_us = window_us;
    if (enable && !g.enabled.load()) {
        g.stop = false;
        g.enabled.store(true, std::memory_order_release);
        for (int i = 0; i < g.n_lanes; ++i) g.lanes.emplace_back(lane_loop);
    } else if (!enable && g.enabled.load()) {
        g.enabled.store(false, std::memory_order_release);   // new callers take the direct path
        g.stop = true;
        lk.unlock();
        g.lane_cv.notify_all();
        for (std::thread& t : g.lanes)
            if (t.joinable()) t.join();
        g.lanes.clear();
        lk.lock();
        // what the open slot still holds is run here; closing its ticket turns later callers away
        Slot* s = g.open.load(std::memory_order_relaxed);
        g.open.store(nullptr, std::memory_order_release);
        const bool wake = g.slot_waiters > 0;
        lk.unlock();
        if (wake) g.slot_cv.notify_all();
        if (s) {
            const uint64_t t = s->ticket.fetch_or(kClosed, std::memory_order_acq_rel);
            if (t_jobs(t) != 0) {
                process_batch(s, t, false, nullptr);
                lk.lock();
                g.batches++;
                g.jobs += t_jobs(t);
                lk.unlock();
            } else {
                release_slot(s);
            }
        }
    }
}

void stats(uint64_t* batches, uint64_t* jobs)
{
    std::lock_guard<std::mutex> lk(g.mu);
    const uint64_t direct = g_direct_jobs.load(std::memory_order_relaxed);
    if (batches) *batches = g.batches + direct;
    if (jobs) *jobs = g.jobs + direct;
}

void lane_profile(uint64_t out[kProfileWords], bool reset)
{
    std::lock_guard<std::mutex> lk(g.mu);
    for (int i = 0; i < kProfileWords; ++i) {
        out[i] = g.prof[i];
        if (reset) g.prof[i] = 0;
    }
}

void shutdown()
{
    set_batching(false, -1, 0);
    std::lock_guard<std::mutex> ctl(g.ctl_mu);
    std::unique_lock<std::mutex> lk(g.mu);
    // every slot must be home before its resources go
    g.slot_waiters++;
    while (g.inited && g.free_slots.size() != (size_t)kSlots) {
        lk.unlock();
        (void)reap_finished_parts();                             // generated bodies nobody is reading any more
        lk.lock();
        if (!g.inited || g.free_slots.size() == (size_t)kSlots) break;
        g.slot_cv.wait_for(lk, std::chrono::milliseconds(1));
    }
    g.slot_waiters--;
    for (Slot& s : g.slots) destroy_slot(s);
    pool_drop_all();
}

// ---------------------------------------------------------------------------
// chunk stream
// ---------------------------------------------------------------------------
struct Stream {
    Slot* s = nullptr;
    size_t chunk = 0;
};

Stream* stream_open(size_t chunk_bytes, std::string* err)
{
    Slot* s = acquire_slot();
    reset_batch(*s);
    if (ensure_slot(*s, 2 * chunk_bytes, 0, 2)) {
        note_failure(*s);
        if (err) *err = s->err;
        release_slot(s);
        return nullptr;
    }
    Stream* st = new Stream;
    st->s = s;
    st->chunk = chunk_bytes;
    return st;
}

uint8_t* stream_buffer(Stream* st, int half) { return st->s->up_host + (size_t)half * st->chunk; }

int stream_submit(Stream* st, int half, const oval_obj_desc& desc)
{
    Slot& s = *st->s;
    const size_t bytes = (size_t)desc.n_units * kUnit;
    oval_obj_desc d = desc;
    d.body_offset = 0;
    char* dev = (char*)s.up_dev + (size_t)half * st->chunk;
    int rc;
    if ((rc = oval_copy_h2d(s.ctx, dev, s.up_host + (size_t)half * st->chunk, bytes))) return slot_fail(s, rc);
    if ((rc = oval_layout_reset(s.ctx, s.val_layout, &d, 1))) return slot_fail(s, rc);
    if ((rc = oval_validate(s.ctx, s.val_layout, dev, bytes, s.res_dev + half, nullptr))) return slot_fail(s, rc);
    if ((rc = oval_copy_d2h(s.ctx, s.res_host + half, s.res_dev + half, sizeof(oval_obj_result)))) return slot_fail(s, rc);
    const hipError_t e = hipEventRecord(s.half_ev[half], oval_internal_ctx_stream(s.ctx));
    if (e != hipSuccess) return hip_fail(s, e, "hipEventRecord");
    return OVAL_OK;
}

int stream_wait(Stream* st, int half, oval_obj_result* res)
{
    Slot& s = *st->s;
    const hipError_t e = hipEventSynchronize(s.half_ev[half]);
    if (e != hipSuccess) return hip_fail(s, e, "hipEventSynchronize");
    *res = s.res_host[half];
    return OVAL_OK;
}

std::string stream_error(Stream* st) { return st->s->err; }

void stream_close(Stream* st)
{
    if (!st) return;
    if (st->s->failed) note_failure(*st->s);
    (void)oval_ctx_sync(st->s->ctx);
    release_slot(st->s);
    delete st;
}

// ---------------------------------------------------------------------------
// generated parts
// ---------------------------------------------------------------------------
namespace {

// Idle pinned buffers.  Pinning memory costs milliseconds per 100 MB, far more
// than generating into it, so buffers are reused; what is kept idle is bounded.
constexpr size_t kPoolKeepBytes = 2ull << 30;
constexpr size_t kPoolKeepBuffers = 16;

struct PinnedBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
};

struct PinnedPool {
    std::mutex mu;
    std::vector<PinnedBuf> idle;
    size_t idle_bytes = 0;
};

PinnedPool g_pool;

// The smallest idle buffer that fits, or a new one (capacity rounded up so that
// bodies of similar size share buffers).
bool pool_take(size_t bytes, PinnedBuf* out, std::string* err)
{
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        size_t best = g_pool.idle.size();
        for (size_t i = 0; i < g_pool.idle.size(); ++i)
            if (g_pool.idle[i].cap >= bytes && (best == g_pool.idle.size() || g_pool.idle[i].cap < g_pool.idle[best].cap))
                best = i;
        if (best != g_pool.idle.size() && g_pool.idle[best].cap <= 2 * bytes + (1u << 20)) {
            *out = g_pool.idle[best];
            g_pool.idle_bytes -= out->cap;
            g_pool.idle.erase(g_pool.idle.begin() + (long)best);
            return true;
        }
    }
    size_t cap = 1u << 20;
    while (cap < bytes) cap += cap < (64u << 20) ? cap : (64u << 20);   // doubles up to 64 MiB, then 64 MiB steps
    void* p = nullptr;
    (void)hipSetDevice(g.device);
    const hipError_t e = hipHostMalloc(&p, cap, hipHostMallocDefault);
    if (e != hipSuccess) {
        if (err) *err = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        return false;
    }
    out->p = static_cast<uint8_t*>(p);
    out->cap = cap;
    return true;
}

void pool_give(const PinnedBuf& b)
{
    if (!b.p) return;
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        if (g_pool.idle.size() < kPoolKeepBuffers && g_pool.idle_bytes + b.cap <= kPoolKeepBytes) {
            g_pool.idle.push_back(b);
            g_pool.idle_bytes += b.cap;
            return;
        }
    }
    (void)hipHostFree(b.p);
}

void pool_drop_all()
{
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (const PinnedBuf& b : g_pool.idle) (void)hipHostFree(b.p);
    g_pool.idle.clear();
    g_pool.idle_bytes = 0;
}

}  // namespace

struct Parts {
    Slot* s = nullptr;                    // held until the last slice has landed
    PinnedBuf buf;
    size_t bytes = 0;
    std::vector<hipEvent_t> landed;       // one per staging slice
    std::mutex mu;
    uint32_t landed_upto = 0;             // slices known to have arrived
    std::string err;
};

namespace {

std::mutex g_parts_mu;
std::vector<Parts*> g_live_parts;          // bodies that still hold a slot

void parts_forget(Parts* p)
{
    std::lock_guard<std::mutex> lk(g_parts_mu);
    for (size_t i = 0; i < g_live_parts.size(); ++i)
        if (g_live_parts[i] == p) {
            g_live_parts[i] = g_live_parts.back();
            g_live_parts.pop_back();
            return;
        }
}

// p->mu held.  Waits for slices [landed_upto, upto) and hands the slot back once
// the last one is in.
bool parts_advance(Parts* p, uint32_t upto)
{
    while (p->landed_upto < upto) {                          // slices may land out of order: every one is waited for
        const hipError_t e = hipEventSynchronize(p->landed[p->landed_upto]);
        if (e != hipSuccess) {
            p->err = std::string("hipEventSynchronize: ") + hipGetErrorString(e);
            return false;
        }
        p->landed_upto++;
    }
    if (p->s && p->landed_upto == p->landed.size()) {
        Slot* s = p->s;
        p->s = nullptr;
        parts_forget(p);
        release_slot(s);
    }
    return true;
}

bool reap_finished_parts()
{
    // candidates first, so that no body's lock is taken while the registry's is held
    std::vector<Parts*> live;
    {
        std::lock_guard<std::mutex> lk(g_parts_mu);
        live = g_live_parts;
    }
    bool any = false;
    for (Parts* p : live) {
        std::unique_lock<std::mutex> reg(g_parts_mu);            // p stays alive while it is registered
        bool still = false;
        for (Parts* q : g_live_parts) still |= q == p;
        if (!still || !p->mu.try_lock()) continue;
        reg.unlock();
        if (p->s && !p->landed.empty() && hipEventQuery(p->landed.back()) == hipSuccess) {
            // slices land out of order across the two copy streams: the last event alone does not say all are in
            bool all = true;
            for (size_t i = p->landed_upto; i < p->landed.size() && all; ++i) all = hipEventQuery(p->landed[i]) == hipSuccess;
            if (all) any |= parts_advance(p, (uint32_t)p->landed.size());
        } else {
            (void)hipGetLastError();                             // hipErrorNotReady is not an error here
        }
        p->mu.unlock();
    }
    return any;
}

}  // namespace

Parts* parts_open(const oval_obj_desc& desc, std::string* err)
{
    Parts* p = new Parts;
    p->bytes = (size_t)desc.n_units * kUnit;
    if (!pool_take(p->bytes ? p->bytes : 1, &p->buf, err)) {
        delete p;
        return nullptr;
    }
    if (p->bytes == 0) return p;
    if (p->bytes <= kSharedJobMax) {
        // small enough to share a launch with other callers' jobs
        Job job;
        job.is_fill = true;
        job.desc = desc;
        job.out = p->buf.p;
        execute(job);
        if (job.failed) {
            if (err) *err = job.err;
            pool_give(p->buf);
            delete p;
            return nullptr;
        }
        return p;
    }
    Slot* s = acquire_slot();
    reset_batch(*s);
    int rc = ensure_slot(*s, 0, 0, 0);
    const uint32_t n_slices = oval_internal_host_slices(p->bytes);
    hipError_t e = hipSuccess;
    if (!rc) {
        p->landed.resize(n_slices, nullptr);
        for (uint32_t i = 0; i < n_slices && e == hipSuccess; ++i)
            e = hipEventCreateWithFlags(&p->landed[i], hipEventDisableTiming);
        if (e != hipSuccess) rc = hip_fail(*s, e, "hipEventCreate");
    }
    if (!rc) {
        oval_obj_desc d = desc;
        d.body_offset = 0;
        rc = oval_internal_fill_host(s->ctx, &d, 1, p->buf.p, p->bytes, p->landed.data(), n_slices, false);
        if (rc) slot_fail(*s, rc);
    }
    if (rc) {
        note_failure(*s);
        if (err) *err = s->err;
        if (s->ctx) (void)oval_ctx_sync(s->ctx);
        release_slot(s);
        for (hipEvent_t ev : p->landed)
            if (ev) (void)hipEventDestroy(ev);
        pool_give(p->buf);
        delete p;
        return nullptr;
    }
    p->s = s;
    {
        std::lock_guard<std::mutex> lk(g_parts_mu);
        g_live_parts.push_back(p);
    }
    return p;
}

const uint8_t* parts_wait(Parts* p, size_t begin, size_t end, std::string* err)
{
    (void)begin;
    if (end > p->bytes) end = p->bytes;
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->landed.empty() && end > 0) {
        const uint64_t slice = oval_internal_host_slice_bytes();
        const uint32_t upto = (uint32_t)((end + slice - 1) / slice);
        if (!parts_advance(p, upto)) {
            if (err) *err = p->err;
            return nullptr;
        }
    }
    return p->buf.p;
}

void parts_close(Parts* p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        // copies still in flight write into the buffer: they must finish before it is reused
        if (!parts_advance(p, (uint32_t)p->landed.size()) && p->s) {
            (void)oval_ctx_sync(p->s->ctx);
            release_slot(p->s);
            p->s = nullptr;
        }
    }
    parts_forget(p);                                         // (a body that never gave its slot back is still listed)
    { std::lock_guard<std::mutex> lk(p->mu); }               // a reaper that found it listed a moment ago is done with it
    for (hipEvent_t ev : p->landed)
        if (ev) (void)hipEventDestroy(ev);
    pool_give(p->buf);
    delete p;
}

void parts_pool_stats(size_t* buffers, size_t* bytes)
{
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (buffers) *buffers = g_pool.idle.size();
    if (bytes) *bytes = g_pool.idle_bytes;
}

}  // namespace engine
}  // namespace oval
