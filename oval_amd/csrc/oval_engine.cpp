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
// A caller need not wait inside the library at all: submit() reserves and writes a job like execute()
// does and returns, try_complete() reads one word of the slot (`completed`, set by the lane to the batch's
// sequence number) and collects the outcome once the batch is over.  Such jobs share launches with the
// blocking callers'; a batch that holds any of them has its sleepers woken by the lane directly (a job
// without a waiting caller cannot pass a wake-up on), a batch of nothing else wakes nobody.  A host
// language with cheap tasks of its own (goroutines that yield between polls) thereby pays neither sleep
// nor wake-up per call.
//
// One engine (everything above) per GPU of the process: a call is routed by its
// worker ID, so a worker's objects always meet the same device, and engines share
// nothing but the host's cores.
//
// Every wait on the device ends: waits poll with a wall-clock limit
// (oval_set_device_timeout_ms), and a slot whose batch did not finish in time is
// set aside for good (its memory may still be written by the device) while the
// other slots go on working.
//
// Large work goes other ways: a batch form (run) beyond 16 MiB travels as
// sub-batches through up to three slots with its host copies on helper threads,
// and a single object of 8 MiB or more takes the sliced host-memory path of the
// C ABI on a slot's context (run_large).
#include "oval_engine.hpp"

#include <hip/hip_runtime.h>
#include <linux/futex.h>
#include <sched.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "oval_capi_internal.h"
#include "oval_kernels.h"

namespace oval {
namespace engine {
namespace {

constexpr size_t kUnit = OVAL_UNIT_SIZE;
// Body capacity of a shared slot, per direction.  A slot that has served shared launches keeps 2 x 16 MiB of pinned
// and as much device memory for the life of the engine; shared use prefers slots that have them already
// (pop_free_locked), so their number follows the batches that were ever in flight or being read out at once: a
// handful under blocking callers, up to all 34 (2.1 GiB pinned + 2.1 GiB on the device, per engine) under hundreds of
// two-step calls whose owners poll slowly.
constexpr size_t kSharedBytes = 16u << 20;
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
// How long a caller polls its wake word before it sleeps on it: 0 = never (the default: on a 16-core host
// polling lost at every thread count from 32 to 255, 10-25 % with the adaptive setting and four pollers,
// because the cores a poller holds are the ones the other callers need to copy their bodies), -1 = as
// long as batches have lately taken on the device (while fewer callers than g_spin_budget are polling),
// > 0 = that many ns.
std::atomic<int> g_spin_ns{0};
std::atomic<int> g_spin_budget{-1};        // callers that may poll at once; -1: a quarter of the host's cores
std::atomic<int> g_spinners{0};
std::atomic<int> g_test_lose_flags{0};     // test hook: that many coming flag batches never see their flag
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

// A wake word counts completions in steps of two; its lowest bit says that the job's caller is asleep on
// it (so whoever completes the job knows whether a system call is needed to wake it).
struct alignas(64) WakeWord {
    std::atomic<uint32_t> v{0};
};

struct Pipeline;

struct Slot {
    Pipeline* pipe = nullptr;             // the engine this slot belongs to
    bool quarantined = false;             // its batch never finished: the device may still write its memory
    bool counted_out = false;             // counted in Pipeline::quarantined (mu held): set aside, or still out when shutdown() gave up on it
    bool lose_flag = false;               // test hook: the batch in flight was given a flag word nobody reads
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
    // Jobs whose callers do not wait inside the library (submit / try_complete): they watch `completed`,
    // which takes the value of `batch_seq` when the batch is over, instead of a wake word of their own.
    uint32_t batch_seq = 0;               // counts the batches of this slot as the open slot; fixed while a batch has readers
    std::atomic<uint32_t> completed{0};
    std::atomic<uint32_t> n_async{0};     // such jobs in the batch being assembled
    std::atomic<uint32_t> written{0};     // callers that have finished writing their descriptor / body
    std::atomic<int> readers{0};          // callers that have not collected their outcome yet
    // One futex word per job (validate jobs first, then fill jobs), each in a
    // cache line of its own: a caller sleeps on its own word, so the sleepers of
    // a batch do not queue on one kernel futex bucket, and is woken by its parent
    // in a tree over the batch's job numbers (see wake_children).
    std::unique_ptr<WakeWord[]> words;
    // One byte per wake word: the job at that place has no caller waiting inside the library (submit()).  Every job
    // writes its own byte before it counts itself in `written`.
    std::unique_ptr<uint8_t[]> no_waiter;
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
    int index = 0;                        // which engine of the process this is
    int quarantined = 0;                  // slots set aside for good (see Slot::quarantined)
    std::atomic<uint64_t> batch_ns{20000};   // how long a shared batch has lately taken from submit to completion (lanes write, callers read)
    // generated bodies that still hold a slot of this engine, and its idle pinned buffers
    std::mutex parts_mu;
    std::vector<struct Parts*> live_parts;
    std::mutex pool_mu;
    std::vector<struct PinnedBuf> pool_idle;
    size_t pool_idle_bytes = 0;
};

struct PinnedBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
};

// Engines are never destroyed: a process that exits while micro-batching is on still has lane threads
// asleep in lane_loop, and destroying their std::thread objects (or the mutex they wait on) at exit would
// abort it.
constexpr int kMaxEngines = 16;
Pipeline* g_engines[kMaxEngines] = {};
std::atomic<int> g_n_engines{0};
std::mutex g_engines_mu;                  // engine table, set_batching / shutdown (held across the joins)
struct Config {                           // what set_lanes / set_batching said, for engines made later
    int n_lanes = 2, blocking = 0, window_us = 0;
    bool enabled = false;
} g_cfg;

void ensure_inited_locked(Pipeline& g);

// The engine table with at least one engine in it (a process that never called init gets one on device 0).
int engines()
{
    int n = g_n_engines.load(std::memory_order_acquire);
    if (n) return n;
    std::lock_guard<std::mutex> lk(g_engines_mu);
    n = g_n_engines.load(std::memory_order_relaxed);
    if (n) return n;
    if (!g_engines[0]) g_engines[0] = new Pipeline;
    g_engines[0]->index = 0;
    g_n_engines.store(1, std::memory_order_release);
    return 1;
}

// Calls are spread by worker ID (internal/runner/runner.go:110-113 stamps consecutive IDs on the workers
// of a process): all objects of a worker meet one device.
Pipeline& engine_for(uint32_t route)
{
    const int n = engines();
    return *g_engines[n == 1 ? 0 : (int)(route % (uint32_t)n)];
}

void pool_drop_all(Pipeline& g);          // pinned buffers of the generated-parts pool (below)

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
        if ((rc = oval_ctx_create(s.pipe->device, nullptr, &s.ctx))) return slot_fail(s, rc);
        if ((rc = oval_layout_create(s.ctx, nullptr, 0, &s.fill_layout))) return slot_fail(s, rc);
        if ((rc = oval_layout_create(s.ctx, nullptr, 0, &s.val_layout))) return slot_fail(s, rc);
        hipError_t e = hipEventCreateWithFlags(&s.done_spin, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done_block, hipEventDisableTiming | hipEventBlockingSync);
        for (int h = 0; h < 2 && e == hipSuccess; ++h) e = hipEventCreateWithFlags(&s.half_ev[h], hipEventDisableTiming);
        if (e != hipSuccess) return hip_fail(s, e, "hipEventCreate");
        if ((rc = oval_internal_pinned_alloc_coherent(s.ctx, 128, (void**)&s.done_flag))) return slot_fail(s, rc);
        if ((rc = oval_dev_alloc(s.ctx, 64, (void**)&s.done_count_dev))) return slot_fail(s, rc);
        s.done_flag[0] = 0;
        s.done_flag[16] = 0;                                     // a second cache line: where a lost flag goes (test hook)
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
    }
    // The wake words belong to shared use, not to a capacity: a slot whose arrays grew for a batch form
    // (any job count) may become the open slot later without growing again.
    if (jobs >= kSharedJobs && !s.words) {
        std::unique_ptr<uint8_t[]> flags(new (std::nothrow) uint8_t[2 * kSharedJobs]());
        std::unique_ptr<WakeWord[]> words(new (std::nothrow) WakeWord[2 * kSharedJobs]);
        if (!flags || !words) {
            s.err = "out of host memory for a staging slot's wake words";
            s.failed = true;
            return OVAL_E_NOMEM;
        }
        s.no_waiter = std::move(flags);
        s.words = std::move(words);
    }
    return OVAL_OK;
}

void destroy_slot(Slot& s)
{
    if (!s.ctx || s.quarantined) return;                         // what the device may still write is never freed
    if (oval_internal_ctx_sync_all(s.ctx)) {
        s.quarantined = true;
        return;
    }
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
    s.n_async.store(0, std::memory_order_relaxed);
    s.failed = false;
    s.err.clear();
}

// Everything a batch needs, enqueued on the slot's stream; no host wait.
int submit(Slot& s, bool blocking)
{
    int rc;
    hipStream_t stream = oval_internal_ctx_stream(s.ctx);
    (void)hipSetDevice(s.pipe->device);                          // a caller's thread may last have used another engine's GPU
    s.by_flag = false;
    s.lose_flag = false;
    if (s.max_job_bytes <= kSharedJobMax && s.val_bytes + s.fill_bytes <= kZeroCopyMax) {
        // (every workgroup pays a system-wide release of its stores for the flag: with a few dozen of them
        // that costs the device more than the event saves the host — 38 jobs: 22.5 against 19.0 us)
        const uint32_t n_jobs = s.n_val + s.n_fill;
        const bool flag = !blocking && g_flag_completion.load(std::memory_order_relaxed) && n_jobs != 0 && n_jobs <= kFlagMaxJobs;
        if (flag) ++s.done_seq;
        if (flag && g_test_lose_flags.load(std::memory_order_relaxed) > 0 &&
            g_test_lose_flags.fetch_sub(1, std::memory_order_relaxed) > 0)
            s.lose_flag = true;                                  // the kernel reports to a word wait_done does not read
        const hipError_t e = launch_host_batch(stream, s.vdescs, s.n_val, s.fdescs, s.n_fill, s.up_host, s.down_host,
                                               s.res_host, flag ? s.done_count_dev : nullptr,
                                               flag ? (s.lose_flag ? s.done_flag + 16 : s.done_flag) : nullptr, s.done_seq);
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

// The slot's batch did not finish within the time limit: whatever it was given (pinned and device
// buffers, result arrays) may still be written by the device, so the slot never carries work again.
int slot_timed_out(Slot& s, const char* what)
{
    s.quarantined = true;
    s.failed = true;
    s.err = std::string("device did not finish the batch (") + what + ") within " +
            std::to_string(oval_internal_device_timeout_ns() / 1000000ull) + " ms; its staging slot is set aside";
    return OVAL_E_HIP;
}

int wait_done(Slot& s, bool blocking)
{
    if (s.by_flag) {
        s.by_flag = false;
        volatile uint32_t* flag = s.done_flag;
        uint64_t t0 = 0;
        for (uint64_t spins = 0; *flag != s.done_seq; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 0x3FFFu) == 0x3FFFu) {
                // every millisecond or so: the time limit, and a stream that has gone idle or bad without the flag
                const uint64_t now = now_ns(), limit = oval_internal_device_timeout_ns();
                if (!t0) t0 = now;
                if (limit && now - t0 > limit) return slot_timed_out(s, "completion flag");
                const hipError_t q = s.lose_flag ? hipErrorNotReady : hipStreamQuery(oval_internal_ctx_stream(s.ctx));
                if (q == hipSuccess) {
                    if (*flag == s.done_seq) break;
                    // the workgroup count of a batch that ended early must not carry over into the next one
                    (void)hipMemset(s.done_count_dev, 0, 64);
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
    (void)oval_internal_wait_timed_out(true);
    const hipError_t e = oval_internal_wait_event(blocking ? s.done_block : s.done_spin, blocking);
    if (e != hipSuccess) {
        if (oval_internal_wait_timed_out(true)) return slot_timed_out(s, "completion event");
        return hip_fail(s, e, "waiting for the batch");
    }
    return OVAL_OK;
}

// After a failure: everything the slot's context still has in flight, with the time limit.
void settle_slot(Slot& s)
{
    if (!s.ctx || s.quarantined) return;
    (void)oval_internal_wait_timed_out(true);
    if (oval_internal_ctx_sync_all(s.ctx) && oval_internal_wait_timed_out(true)) s.quarantined = true;
}

std::mutex g_err_mu;
std::string g_last_err;                   // text of the most recent engine failure, whichever engine

void note_failure(const Slot& s)
{
    {
        std::lock_guard<std::mutex> lk(s.pipe->mu);
        s.pipe->err = s.err;
    }
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_last_err = s.err;
}

void ensure_inited_locked(Pipeline& g)
{
    if (g.inited) return;
    for (Slot& s : g.slots) {
        s.pipe = &g;
        g.free_slots.push_back(&s);
    }
    g.inited = true;
}

bool shared_ready(const Slot& s) { return s.ctx && s.up_cap >= kSharedBytes && s.down_cap >= kSharedBytes && s.job_cap >= kSharedJobs && s.words; }

// Takes a slot off the free list (mu held, list not empty).  Shared use wants one whose 2 x 16 MiB of
// pinned and device memory exist already; exclusive use leaves those to the micro-batcher when it can,
// so that only a handful of slots ever grow to that size.
Slot* pop_free_locked(Pipeline& g, bool for_shared)
{
    size_t pick = g.free_slots.size() - 1;
    for (size_t i = g.free_slots.size(); i-- > 0;)
        if (shared_ready(*g.free_slots[i]) == for_shared) {
            pick = i;
            break;
        }
    Slot* s = g.free_slots[pick];
    g.free_slots.erase(g.free_slots.begin() + (long)pick);
    return s;
}

// Generated bodies (parts_open) keep their slot until their last slice has landed, but nobody is told
// when that happens: whoever finds no slot free looks whether one of them is done (defined below).
bool reap_finished_parts(Pipeline& g);

// A free slot for the caller's exclusive use; NULL once every slot of the engine has been lost to a
// device that does not answer.
Slot* acquire_slot(Pipeline& g)
{
    std::unique_lock<std::mutex> lk(g.mu);
    ensure_inited_locked(g);
    while (g.free_slots.empty()) {
        if (g.quarantined == kSlots) return nullptr;
        lk.unlock();
        const bool reaped = reap_finished_parts(g);
        lk.lock();
        if (reaped || !g.free_slots.empty()) continue;
        g.slot_waiters++;
        g.slot_cv.wait_for(lk, std::chrono::microseconds(200));   // a landed body frees its slot without a notify
        g.slot_waiters--;
    }
    return pop_free_locked(g, false);
}

// A free slot if there is one right now.
Slot* try_acquire_slot(Pipeline& g)
{
    std::lock_guard<std::mutex> lk(g.mu);
    ensure_inited_locked(g);
    if (g.free_slots.empty()) return nullptr;
    return pop_free_locked(g, false);
}

const char* const kNoSlots = "every staging slot of this engine has been set aside after device time-outs";

// Makes `c` (off the free list, owned by the caller) the open slot; `lk` holds mu on entry and on return.
// The first shared use of a slot allocates 2 x 16 MiB of pinned and as much device memory, which takes
// milliseconds: mu is dropped for that, and if somebody else has installed an open slot meanwhile (or
// batching has ended) `c` is not installed.  False: `c` is still the caller's.
bool install_open(std::unique_lock<std::mutex>& lk, Pipeline& g, Slot* c)
{
    c->failed = false;                                           // (callers tell "could not allocate" from "not needed" by this)
    if (!shared_ready(*c)) {
        lk.unlock();
        const int rc = ensure_slot(*c, kSharedBytes, kSharedBytes, kSharedJobs);
        lk.lock();
        if (rc) {
            g.err = c->err;
            return false;
        }
        if (g.open.load(std::memory_order_relaxed) || !g.enabled.load(std::memory_order_relaxed) || g.stop) return false;
    }
    reset_batch(*c);
    c->batch_seq++;
    c->ticket.store(0, std::memory_order_release);               // (a caller still holding this slot's address from its last time open may reserve through the ticket alone)
    g.open.store(c, std::memory_order_release);
    return true;
}

void release_slot(Slot* s)
{
    Pipeline& g = *s->pipe;
    bool wake;
    {
        std::unique_lock<std::mutex> lk(g.mu);
        if (s->quarantined) {
            if (!s->counted_out) g.quarantined++;                // never on the free list again
            s->counted_out = true;
        } else if (s->counted_out) {
            // shutdown() gave up waiting for this slot (jobs nobody had collected yet): it comes home now
            s->counted_out = false;
            g.quarantined--;
            g.free_slots.push_back(s);
        } else if (!(g.enabled.load(std::memory_order_relaxed) && !g.stop && !g.open.load(std::memory_order_relaxed) &&
                     shared_ready(*s) && install_open(lk, g, s))) {
            // (callers may be waiting for an open slot because every slot was busy: a slot that is ready for
            // shared use becomes the open one at once)
            g.free_slots.push_back(s);
        }
        wake = g.slot_waiters > 0;
    }
    if (wake) g.slot_cv.notify_all();
}

// Completes a wake word: the next even value, and a system call only if its caller sleeps on it.
void complete_word(std::atomic<uint32_t>* w)
{
    uint32_t old = w->load(std::memory_order_relaxed);
    while (!w->compare_exchange_weak(old, (old | 1u) + 1u, std::memory_order_release, std::memory_order_relaxed)) {
    }
    if (old & 1u) futex_wake(w, 1);
}

void wake_job(Slot* s, uint32_t d)
{
    complete_word(&s->words[d < s->n_val ? d : kSharedJobs + (d - s->n_val)].v);
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
    for (uint32_t c = 0; c < roots; ++c) complete_word(&words[c < n_val ? c : kSharedJobs + (c - n_val)].v);
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
    else settle_slot(*s);                                        // whatever was enqueued before the failure
    const uint64_t t3 = now_ns();
    if (rc) note_failure(*s);
    else s->pipe->batch_ns.store((3 * s->pipe->batch_ns.load(std::memory_order_relaxed) + (t3 - t1)) / 4,
                                 std::memory_order_relaxed);     // (two lanes may each lose the other's sample: it is an estimate)
    s->readers.store((int)n_jobs, std::memory_order_relaxed);
    // Jobs submitted without a waiting caller learn of the end from `completed`.  Nobody wakes on their
    // behalf, so they cannot be links of the wake tree: a batch that holds any of them has every sleeping
    // caller woken from here (all jobs are roots), a batch of nothing else needs no wake-up at all.
    const uint32_t n_async = s->n_async.load(std::memory_order_acquire);
    if (n_async != 0) {
        // Only the words of jobs with a waiting caller are touched, and the walk ends with the last of them: once that
        // caller and the pollers are through, the slot may be open again and its words and bytes belong to new jobs.
        s->wake_fanout = 0;
        const uint32_t n_val = s->n_val, n_waiting = n_jobs - n_async;
        WakeWord* const words = s->words.get();
        const uint8_t* const no_waiter = s->no_waiter.get();
        s->completed.store(s->batch_seq, std::memory_order_release);
        for (uint32_t c = 0, woken = 0; woken < n_waiting && c < n_jobs; ++c) {
            const uint32_t wi = c < n_val ? c : (uint32_t)kSharedJobs + (c - n_val);
            if (no_waiter[wi]) continue;
            ++woken;
            complete_word(&words[wi].v);
        }
        if (phase_ns) { phase_ns[0] = t1 - t0; phase_ns[1] = t2 - t1; phase_ns[2] = t3 - t2; phase_ns[3] = now_ns() - t3; }
        return;
    }
    wake_roots(s);
    if (phase_ns) {
        phase_ns[0] = t1 - t0; phase_ns[1] = t2 - t1; phase_ns[2] = t3 - t2; phase_ns[3] = now_ns() - t3;
    }
}

void lane_loop(Pipeline* pipe)
{
    Pipeline& g = *pipe;
    std::unique_lock<std::mutex> lk(g.mu);
    (void)hipSetDevice(g.device);                                // a new thread starts on device 0
    (void)prctl(PR_SET_TIMERSLACK, 1000ul, 0ul, 0ul, 0ul);        // a lane that naps while its batch is on the device naps for microseconds, not for the default 50
    for (;;) {
        g.lane_cv.wait(lk, [&g] {
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
            Slot* f = pop_free_locked(g, true);
            if (!install_open(lk, g, f)) g.free_slots.push_back(f);
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

namespace {
void start_lanes_locked(Pipeline& g);     // (g_engines_mu held) below, with set_batching

bool engine_idle(Pipeline& g)
{
    std::lock_guard<std::mutex> lk(g.mu);
    ensure_inited_locked(g);
    for (const Slot& s : g.slots)
        if (s.ctx && !s.quarantined) return false;               // (slots set aside for good keep what they hold)
    return true;
}
}  // namespace

int init_engines(const int* devices, int n)
{
    if (n < 1 || n > kMaxEngines || !devices) return oval_internal_fail(OVAL_E_INVALID, "1 to %d engines", kMaxEngines);
    {
        std::lock_guard<std::mutex> lk(g_engines_mu);
        // the first call decides: engines that hold device resources stay as they are
        bool fresh = true;
        const int have = g_n_engines.load(std::memory_order_relaxed);
        for (int k = 0; k < have; ++k) fresh &= engine_idle(*g_engines[k]);
        if (fresh && !g_cfg.enabled) {
            for (int k = 0; k < n; ++k) {
                if (!g_engines[k]) g_engines[k] = new Pipeline;
                g_engines[k]->index = k;
                g_engines[k]->device = devices[k];
                std::lock_guard<std::mutex> elk(g_engines[k]->mu);
                ensure_inited_locked(*g_engines[k]);
            }
            g_n_engines.store(n, std::memory_order_release);
        }
    }
    // one slot's context per engine proves its device is usable
    const int have = engines();
    for (int k = 0; k < have; ++k) {
        Slot* s = acquire_slot(*g_engines[k]);
        if (!s) return oval_internal_fail(OVAL_E_HIP, "%s", kNoSlots);
        const int rc = ensure_slot(*s, 0, 0, 0);
        if (rc) note_failure(*s);
        release_slot(s);
        if (rc) return rc;
    }
    return OVAL_OK;
}

int init(int device)
{
    // OVAL_HIP_ENGINES=N: N engines on devices device, device + 1, ... (modulo the visible GPUs, so that
    // a one-GPU box can run two engines on its one device); "all": one per visible GPU
    int n = 1, n_dev = 1;
    if (const char* e = std::getenv("OVAL_HIP_ENGINES")) {
        (void)oval_device_count(&n_dev);
        if (n_dev < 1) n_dev = 1;
        n = std::strcmp(e, "all") == 0 ? n_dev : std::atoi(e);
        if (n < 1) n = 1;
        if (n > kMaxEngines) n = kMaxEngines;
    }
    int devices[kMaxEngines];
    for (int k = 0; k < n; ++k) devices[k] = n == 1 ? device : (device + k) % n_dev;
    return init_engines(devices, n);
}

int engine_count() { return engines(); }

std::string last_error()
{
    std::lock_guard<std::mutex> lk(g_err_mu);
    return g_last_err;
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
    if (b.rc) settle_slot(*s);                                   // whatever was enqueued before the failure
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

namespace {
void run_on(Pipeline& g, Job* const* jobs, size_t n);
}

void run(Job* const* jobs, size_t n)
{
    if (n == 0) return;
    const int n_eng = engines();
    if (n_eng == 1 || n == 1) {
        run_on(engine_for(jobs[0]->desc.worker_id), jobs, n);
        return;
    }
    // one list per engine (a worker's objects meet one device), each list through its own engine; the
    // lists run side by side so that the GPUs and their links work at the same time
    std::vector<std::vector<Job*>> lists((size_t)n_eng);
    for (size_t i = 0; i < n; ++i) lists[jobs[i]->desc.worker_id % (uint32_t)n_eng].push_back(jobs[i]);
    std::vector<std::thread> side;
    int mine = -1;
    for (int k = 0; k < n_eng; ++k) {
        if (lists[(size_t)k].empty()) continue;
        if (mine < 0) { mine = k; continue; }
        side.emplace_back([&lists, k] { run_on(*g_engines[k], lists[(size_t)k].data(), lists[(size_t)k].size()); });
    }
    if (mine >= 0) run_on(*g_engines[mine], lists[(size_t)mine].data(), lists[(size_t)mine].size());
    for (std::thread& t : side) t.join();
}

namespace {
void fail_jobs(Job* const* jobs, size_t n, const char* why)
{
    for (size_t i = 0; i < n; ++i) { jobs[i]->failed = true; jobs[i]->err = why; }
}

void run_on(Pipeline& g, Job* const* jobs, size_t n)
{
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
        Slot* slot = live < depth ? try_acquire_slot(g) : nullptr;
        while (!slot && live) {
            sub_finish(ring[head]);
            head = (head + 1) % depth;
            --live;
            slot = try_acquire_slot(g);
        }
        if (!slot) slot = acquire_slot(g);
        if (!slot) {
            fail_jobs(jobs + first, n - first, kNoSlots);
            break;
        }
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
}  // namespace

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
    Slot* s = acquire_slot(engine_for(job.desc.worker_id));
    if (!s) {
        job.failed = true;
        job.err = kNoSlots;
        return;
    }
    reset_batch(*s);
    int rc = ensure_slot(*s, 0, 0, 0);
    if (!rc) {
        oval_obj_desc d = job.desc;
        d.body_offset = 0;
        const uint64_t bytes = (uint64_t)d.n_units * kUnit;
        (void)hipSetDevice(s->pipe->device);
        (void)oval_internal_wait_timed_out(true);
        if (job.is_fill) rc = oval_fill_batch_host(s->ctx, &d, 1, job.out, bytes);
        else rc = oval_validate_batch_host(s->ctx, &d, 1, job.in, bytes, &job.res, nullptr);
        if (rc) {
            slot_fail(*s, rc);
            if (oval_internal_wait_timed_out(true)) s->quarantined = true;   // copies or kernels of it may still be under way
        }
    }
    if (rc) {
        note_failure(*s);
        job.failed = true;
        job.err = s->err;
        settle_slot(*s);
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
    Pipeline& g = engine_for(job.desc.worker_id);
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
                Slot* f = pop_free_locked(g, true);
                if (!install_open(lk, g, f)) {
                    g.free_slots.push_back(f);
                    if (f->failed) {                             // its buffers could not be allocated
                        job.failed = true;
                        job.err = f->err;
                        return;
                    }
                }
                continue;                                        // installed, or somebody else was quicker
            }
            if (g.quarantined == kSlots) {
                job.failed = true;
                job.err = kNoSlots;
                return;
            }
        } else if (o != c || !full) {
            continue;                                            // it changed while the lock was taken
        }
        lk.unlock();
        if (reap_finished_parts(g)) continue;                    // a landed body was still holding a slot
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
    s->no_waiter[job.is_fill ? kSharedJobs + idx : idx] = 0;
    s->written.fetch_add(1, std::memory_order_release);
    if (t_jobs(before) == 0) {
        // first job of the batch: a lane may be asleep (taking the lock orders
        // this against a lane that has just found the slot empty).  Done after the
        // job is written: between reservation and that point a caller holds up
        // its whole batch, so nothing that can sleep or be preempted belongs there.
        { std::lock_guard<std::mutex> lk(g.mu); }
        g.lane_cv.notify_one();
    }
    // Polling first, while few enough callers do: a batch is back within tens of microseconds, and a caller
    // that sees its word change needs neither a sleep nor a wake-up through the kernel.  Then asleep on the
    // word with its lowest bit set, which tells whoever completes it that a wake-up call is needed.
    int spin_ns = g_spin_ns.load(std::memory_order_relaxed);
    if (spin_ns < 0) {
        const uint64_t twice = 2 * g.batch_ns.load(std::memory_order_relaxed);
        spin_ns = (int)(twice < 200000 ? twice : 200000);
    }
    if (spin_ns > 0) {
        int budget = g_spin_budget.load(std::memory_order_relaxed);
        if (budget < 0) budget = direct_max() / 4;
        if (g_spinners.fetch_add(1, std::memory_order_relaxed) < budget) {
            const uint64_t until = now_ns() + (uint64_t)spin_ns;
            for (uint32_t k = 0; word->load(std::memory_order_acquire) == seq; ++k) {
                __builtin_ia32_pause();
                if ((k & 31u) == 31u && now_ns() >= until) break;
            }
        }
        g_spinners.fetch_sub(1, std::memory_order_relaxed);
    }
    if (word->load(std::memory_order_acquire) == seq) {
        uint32_t expect = seq;
        if (word->compare_exchange_strong(expect, seq | 1u, std::memory_order_acq_rel, std::memory_order_acquire))
            while (word->load(std::memory_order_acquire) == (seq | 1u)) futex_wait(word, seq | 1u);
    }
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

// ---------------------------------------------------------------------------
// jobs whose callers do not wait inside the library
// ---------------------------------------------------------------------------
// A host language with cheap tasks of its own (goroutines) loses more to the two sleeps and two wake-ups
// of a blocking call than the device takes for the work: submit() reserves and writes the job like
// execute() does and returns; try_complete() says whether the batch is over and, once, collects the
// outcome.  The job shares its launch with whatever else arrived, blocking callers included.
int submit(Job& job, Ticket& t)
{
    t.slot = nullptr;
    const uint64_t units = job.desc.n_units;
    const size_t bytes = (size_t)units * kUnit;
    if (bytes == 0 || bytes > kSharedJobMax) return -1;
    Pipeline& g = engine_for(job.desc.worker_id);
    if (!g.enabled.load(std::memory_order_acquire)) return -1;
    const int ushift = job.is_fill ? kFillUnitsShift : kValUnitsShift;
    const int jshift = job.is_fill ? kFillJobsShift : kValJobsShift;
    Slot* s = nullptr;
    uint64_t before = 0;
    // The open slot as it is right now, or none.  A caller that does not wait inside the library is never
    // made to wait for a slot here: it may itself hold jobs that keep slots busy until it collects them.
    for (int attempt = 0; attempt < 4 && !s; ++attempt) {
        Slot* c = g.open.load(std::memory_order_acquire);
        if (!c) {
            std::unique_lock<std::mutex> lk(g.mu);
            if (!g.enabled.load(std::memory_order_relaxed)) return -1;
            if (!g.open.load(std::memory_order_relaxed)) {
                if (g.free_slots.empty()) return g.quarantined == kSlots ? -1 : 0;
                Slot* f = pop_free_locked(g, true);
                if (!install_open(lk, g, f)) {
                    g.free_slots.push_back(f);
                    if (f->failed) return -1;
                }
            }
            continue;
        }
        uint64_t tk = c->ticket.load(std::memory_order_acquire);
        while (!(tk & kClosed)) {
            if (((tk >> ushift) & kUnitsMask) + units > kSharedUnits || ((tk >> jshift) & kJobsMask) + 1 > kSharedJobs) return 0;
            if (c->ticket.compare_exchange_weak(tk, tk + (units << ushift) + (1ull << jshift), std::memory_order_acq_rel,
                                                std::memory_order_acquire)) {
                s = c;
                before = tk;
                break;
            }
        }
    }
    if (!s) return 0;
    const uint32_t idx = (uint32_t)((before >> jshift) & kJobsMask);
    const size_t off = (size_t)((before >> ushift) & kUnitsMask) * kUnit;
    t.slot = s;
    t.batch_seq = s->batch_seq;                                  // fixed from the reservation until this job has been collected
    t.idx = idx;
    t.off = off;
    job.desc.body_offset = off;
    (job.is_fill ? s->fdescs : s->vdescs)[idx] = job.desc;
    if (!job.is_fill) std::memcpy(s->up_host + off, job.in, bytes);
    s->no_waiter[job.is_fill ? kSharedJobs + idx : idx] = 1;
    s->n_async.fetch_add(1, std::memory_order_release);
    s->written.fetch_add(1, std::memory_order_release);
    if (t_jobs(before) == 0) {
        { std::lock_guard<std::mutex> lk(g.mu); }                // see execute(): orders this against a lane about to sleep
        g.lane_cv.notify_one();
    }
    return 1;
}

bool try_complete(Ticket& t, Job& job)
{
    Slot* s = static_cast<Slot*>(t.slot);
    if (s->completed.load(std::memory_order_acquire) != t.batch_seq) return false;
    if (s->failed) {
        job.failed = true;
        job.err = s->err;
    } else if (job.is_fill) {
        std::memcpy(job.out, s->down_host + t.off, (size_t)job.desc.n_units * kUnit);
    } else {
        job.res = s->res_host[t.idx];
    }
    t.slot = nullptr;
    if (s->readers.fetch_sub(1, std::memory_order_acq_rel) == 1) release_slot(s);
    return true;
}

// Where every engine stands, for a tool that has found its calls stuck: written without taking any lock
// (the locks may be what is stuck), so the values are a snapshot that may be torn.
void debug_dump(FILE* out)
{
    const int n = g_n_engines.load(std::memory_order_acquire);
    std::fprintf(out, "engines %d, callers inside execute %d\n", n, g_callers.load());
    for (int k = 0; k < n; ++k) {
        Pipeline& g = *g_engines[k];
        Slot* open = g.open.load(std::memory_order_acquire);
        std::fprintf(out, "engine %d: device %d enabled %d stop %d lanes %zu free %zu quarantined %d slot_waiters %d open %ld\n", k,
                     g.device, (int)g.enabled.load(), (int)g.stop, g.lanes.size(), g.free_slots.size(), g.quarantined,
                     g.slot_waiters, open ? (long)(open - g.slots) : -1L);
        for (int i = 0; i < kSlots; ++i) {
            Slot& s = g.slots[i];
            const uint64_t t = s.ticket.load(std::memory_order_acquire);
            bool is_free = false;
            for (Slot* f : g.free_slots) is_free |= f == &s;
            if (is_free && !(s.readers.load() || t_jobs(t))) continue;
            std::fprintf(out, "  slot %2d%s: ticket %s jobs %u+%u units %u+%u | written %u readers %d n_async %u batch_seq %u completed %u "
                              "by_flag %d failed %d n_val %u n_fill %u\n",
                         i, is_free ? " (free)" : "", (t & kClosed) ? "closed" : "open", t_val_jobs(t), t_fill_jobs(t),
                         t_val_units(t), t_fill_units(t), s.written.load(), s.readers.load(), s.n_async.load(), s.batch_seq,
                         s.completed.load(), (int)s.by_flag, (int)s.failed, s.n_val, s.n_fill);
        }
    }
    std::fflush(out);
}

void set_lanes(int lanes, int blocking)
{
    std::lock_guard<std::mutex> lk(g_engines_mu);
    if (lanes >= 1) g_cfg.n_lanes = lanes > kMaxLanes ? kMaxLanes : lanes;
    if (blocking >= 0) g_cfg.blocking = blocking ? 1 : 0;
}

void set_run_shape(size_t chunk_bytes, int depth)
{
    if (chunk_bytes >= (64u << 10)) g_run_chunk_bytes.store(chunk_bytes, std::memory_order_relaxed);
    if (depth >= 1) g_run_depth.store((size_t)depth > kRunDepthMax ? kRunDepthMax : (size_t)depth, std::memory_order_relaxed);
}

void set_flag_completion(bool on) { g_flag_completion.store(on, std::memory_order_relaxed); }

void set_caller_spin(int ns) { g_spin_ns.store(ns < -1 ? -1 : ns, std::memory_order_relaxed); }

void set_spin_budget(int callers) { g_spin_budget.store(callers < -1 ? -1 : callers, std::memory_order_relaxed); }

void test_lose_flags(int n) { g_test_lose_flags.store(n < 0 ? 0 : n, std::memory_order_relaxed); }

int quarantined_slots()
{
    int q = 0;
    const int n = engines();
    for (int k = 0; k < n; ++k) {
        std::lock_guard<std::mutex> lk(g_engines[k]->mu);
        q += g_engines[k]->quarantined;
    }
    return q;
}

void set_wake_tree(int fanout, int roots)
{
    if (fanout >= 1) g_wake_fanout.store(fanout > 64 ? 64 : fanout, std::memory_order_relaxed);
    if (roots >= 1) g_wake_roots.store(roots > 64 ? 64 : roots, std::memory_order_relaxed);
}

namespace {

// g_engines_mu held.
void start_lanes_locked(Pipeline& g)
{
    std::unique_lock<std::mutex> lk(g.mu);
    ensure_inited_locked(g);
    g.window_us = g_cfg.window_us;
    g.n_lanes = g_cfg.n_lanes;
    g.blocking = g_cfg.blocking;
    if (g.enabled.load()) return;
    g.stop = false;
    // (a lane is an OS thread: the system may refuse one.  With at least one lane the engine batches; with none it
    // stays off and calls go on launching on their own.)
    try {
        for (int i = 0; i < g.n_lanes; ++i) g.lanes.emplace_back(lane_loop, &g);
    } catch (const std::exception& e) {
        g.err = std::string("could not start a lane thread: ") + e.what();
    }
    if (!g.lanes.empty()) g.enabled.store(true, std::memory_order_release);
}

// g_engines_mu held.
void stop_lanes_locked(Pipeline& g)
{
    std::unique_lock<std::mutex> lk(g.mu);
    if (!g.enabled.load()) return;
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

}  // namespace

void set_batching(bool enable, int window_us, size_t /*max_bytes*/)
{
    const int n = engines();
    std::lock_guard<std::mutex> ctl(g_engines_mu);
    if (window_us >= 0) g_cfg.window_us = window_us;
    g_cfg.enabled = enable;
    for (int k = 0; k < n; ++k) {
        if (enable) {
            start_lanes_locked(*g_engines[k]);
        } else {
            {
                std::lock_guard<std::mutex> lk(g_engines[k]->mu);
                g_engines[k]->window_us = g_cfg.window_us;
            }
            stop_lanes_locked(*g_engines[k]);
        }
    }
}

void stats(uint64_t* batches, uint64_t* jobs)
{
    uint64_t b = g_direct_jobs.load(std::memory_order_relaxed), j = b;
    const int n = engines();
    for (int k = 0; k < n; ++k) {
        std::lock_guard<std::mutex> lk(g_engines[k]->mu);
        b += g_engines[k]->batches;
        j += g_engines[k]->jobs;
    }
    if (batches) *batches = b;
    if (jobs) *jobs = j;
}

void lane_profile(uint64_t out[kProfileWords], bool reset)
{
    for (int i = 0; i < kProfileWords; ++i) out[i] = 0;
    const int n = engines();
    for (int k = 0; k < n; ++k) {
        Pipeline& g = *g_engines[k];
        std::lock_guard<std::mutex> lk(g.mu);
        for (int i = 0; i < kProfileWords; ++i) {
            out[i] += g.prof[i];
            if (reset) g.prof[i] = 0;
        }
    }
}

void shutdown()
{
    set_batching(false, -1, 0);
    const int n = engines();
    std::lock_guard<std::mutex> ctl(g_engines_mu);
    for (int k = 0; k < n; ++k) {
        Pipeline& g = *g_engines[k];
        std::unique_lock<std::mutex> lk(g.mu);
        // Every slot must be home (or set aside for good) before its resources go.  A slot can stay out for
        // ever (a call in two steps that nobody polls to its end, a generated body nobody closes): after the
        // device time limit (at least a second) such a slot keeps what it holds and is left where it is; it
        // rejoins the free list if its holder lets go of it later.
        const uint64_t limit = oval_internal_device_timeout_ns();
        const uint64_t give_up = now_ns() + (limit > 1000000000ull ? limit : 1000000000ull);
        g.slot_waiters++;
        while (g.inited && g.free_slots.size() + (size_t)g.quarantined != (size_t)kSlots) {
            lk.unlock();
            (void)reap_finished_parts(g);                        // generated bodies nobody is reading any more
            lk.lock();
            if (!g.inited || g.free_slots.size() + (size_t)g.quarantined == (size_t)kSlots) break;
            if (now_ns() > give_up) {
                for (Slot& s : g.slots) {
                    bool home = s.counted_out;
                    for (const Slot* f : g.free_slots) home |= f == &s;
                    if (!home) {
                        s.counted_out = true;
                        g.quarantined++;
                    }
                }
                break;
            }
            g.slot_cv.wait_for(lk, std::chrono::milliseconds(1));
        }
        g.slot_waiters--;
        for (Slot& s : g.slots)
            if (!s.counted_out) destroy_slot(s);                 // (a slot still out is in use, one set aside may still be written: both keep what they hold)
        lk.unlock();
        pool_drop_all(g);
    }
}

// ---------------------------------------------------------------------------
// chunk stream
// ---------------------------------------------------------------------------
struct Stream {
    Slot* s = nullptr;
    size_t chunk = 0;
};

Stream* stream_open(size_t chunk_bytes, uint32_t route, std::string* err)
{
    Slot* s = acquire_slot(engine_for(route));
    if (!s) {
        if (err) *err = kNoSlots;
        return nullptr;
    }
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
    (void)hipSetDevice(s.pipe->device);
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
    (void)oval_internal_wait_timed_out(true);
    const hipError_t e = oval_internal_wait_event(s.half_ev[half], false);
    if (e != hipSuccess) {
        if (oval_internal_wait_timed_out(true)) return slot_timed_out(s, "a chunk of a streamed body");
        return hip_fail(s, e, "waiting for a chunk");
    }
    *res = s.res_host[half];
    return OVAL_OK;
}

std::string stream_error(Stream* st) { return st->s->err; }

void stream_close(Stream* st)
{
    if (!st) return;
    if (st->s->failed) note_failure(*st->s);
    settle_slot(*st->s);
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

// The smallest idle buffer that fits, or a new one (capacity rounded up so that
// bodies of similar size share buffers).
bool pool_take(Pipeline& g, size_t bytes, PinnedBuf* out, std::string* err)
{
    {
        std::lock_guard<std::mutex> lk(g.pool_mu);
        std::vector<PinnedBuf>& idle = g.pool_idle;
        size_t best = idle.size();
        for (size_t i = 0; i < idle.size(); ++i)
            if (idle[i].cap >= bytes && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
        if (best != idle.size() && idle[best].cap <= 2 * bytes + (1u << 20)) {
            *out = idle[best];
            g.pool_idle_bytes -= out->cap;
            idle.erase(idle.begin() + (long)best);
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

void pool_give(Pipeline& g, const PinnedBuf& b)
{
    if (!b.p) return;
    {
        std::lock_guard<std::mutex> lk(g.pool_mu);
        if (g.pool_idle.size() < kPoolKeepBuffers && g.pool_idle_bytes + b.cap <= kPoolKeepBytes) {
            g.pool_idle.push_back(b);
            g.pool_idle_bytes += b.cap;
            return;
        }
    }
    (void)hipHostFree(b.p);
}

void pool_drop_all(Pipeline& g)
{
    std::lock_guard<std::mutex> lk(g.pool_mu);
    for (const PinnedBuf& b : g.pool_idle) (void)hipHostFree(b.p);
    g.pool_idle.clear();
    g.pool_idle_bytes = 0;
}

}  // namespace

struct Parts {
    Pipeline* pipe = nullptr;             // the engine whose pool and slots the body uses
    Slot* s = nullptr;                    // held until the last slice has landed
    PinnedBuf buf;
    size_t bytes = 0;
    std::vector<hipEvent_t> landed;       // one per staging slice
    std::mutex mu;
    uint32_t landed_upto = 0;             // slices known to have arrived
    bool timed_out = false;               // a slice did not land within the time limit: buffer and slot are never reused
    std::string err;
};

namespace {

void parts_forget(Parts* p)
{
    std::lock_guard<std::mutex> lk(p->pipe->parts_mu);
    std::vector<Parts*>& live = p->pipe->live_parts;
    for (size_t i = 0; i < live.size(); ++i)
        if (live[i] == p) {
            live[i] = live.back();
            live.pop_back();
            return;
        }
}

// p->mu held.  Waits for slices [landed_upto, upto) and hands the slot back once
// the last one is in.
bool parts_advance(Parts* p, uint32_t upto)
{
    while (p->landed_upto < upto) {                          // slices may land out of order: every one is waited for
        (void)oval_internal_wait_timed_out(true);
        const hipError_t e = oval_internal_wait_event(p->landed[p->landed_upto], false);
        if (e != hipSuccess) {
            p->err = std::string("waiting for a slice of a generated body: ") + oval_internal_wait_error(e);
            if (oval_internal_wait_timed_out(true)) p->timed_out = true;
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

bool reap_finished_parts(Pipeline& g)
{
    // candidates first, so that no body's lock is taken while the registry's is held
    std::vector<Parts*> live;
    {
        std::lock_guard<std::mutex> lk(g.parts_mu);
        live = g.live_parts;
    }
    bool any = false;
    for (Parts* p : live) {
        std::unique_lock<std::mutex> reg(g.parts_mu);            // p stays alive while it is registered
        bool still = false;
        for (Parts* q : g.live_parts) still |= q == p;
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
    Pipeline& g = engine_for(desc.worker_id);
    Parts* p = new Parts;
    p->pipe = &g;
    p->bytes = (size_t)desc.n_units * kUnit;
    if (!pool_take(g, p->bytes ? p->bytes : 1, &p->buf, err)) {
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
            // (a body that failed by a time-out may still be written by the device: its buffer is not reused)
            if (job.err.find("set aside") == std::string::npos) pool_give(g, p->buf);
            delete p;
            return nullptr;
        }
        return p;
    }
    Slot* s = acquire_slot(g);
    if (!s) {
        if (err) *err = kNoSlots;
        pool_give(g, p->buf);
        delete p;
        return nullptr;
    }
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
        (void)hipSetDevice(g.device);
        (void)oval_internal_wait_timed_out(true);
        rc = oval_internal_fill_host(s->ctx, &d, 1, p->buf.p, p->bytes, p->landed.data(), n_slices, false);
        if (rc) {
            slot_fail(*s, rc);
            if (oval_internal_wait_timed_out(true)) s->quarantined = true;
        }
    }
    if (rc) {
        note_failure(*s);
        if (err) *err = s->err;
        settle_slot(*s);                                         // slice copies on all three streams
        const bool lost = s->quarantined;
        release_slot(s);
        for (hipEvent_t ev : p->landed)
            if (ev) (void)hipEventDestroy(ev);
        if (!lost) pool_give(g, p->buf);                         // (a buffer the device may still write is leaked, not pooled)
        delete p;
        return nullptr;
    }
    p->s = s;
    {
        std::lock_guard<std::mutex> lk(g.parts_mu);
        g.live_parts.push_back(p);
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
    bool lost = false;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        // Copies still in flight write into the buffer: they must finish before it is reused.  The slices
        // travel on the context's two copy streams, so after a failed wait all three streams are waited
        // for (with the time limit); a body whose copies cannot be seen to end keeps its buffer and its
        // slot out of circulation for good.
        if (!parts_advance(p, (uint32_t)p->landed.size())) {
            lost = p->timed_out;
            if (p->s) {
                if (lost) p->s->quarantined = true;
                else settle_slot(*p->s);
                lost |= p->s->quarantined;
                release_slot(p->s);
                p->s = nullptr;
            }
        }
    }
    parts_forget(p);                                         // (a body that never gave its slot back is still listed)
    { std::lock_guard<std::mutex> lk(p->mu); }               // a reaper that found it listed a moment ago is done with it
    for (hipEvent_t ev : p->landed)
        if (ev) (void)hipEventDestroy(ev);
    if (!lost) pool_give(*p->pipe, p->buf);
    delete p;
}

void parts_pool_stats(size_t* buffers, size_t* bytes)
{
    size_t nb = 0, by = 0;
    const int n = engines();
    for (int k = 0; k < n; ++k) {
        std::lock_guard<std::mutex> lk(g_engines[k]->pool_mu);
        nb += g_engines[k]->pool_idle.size();
        by += g_engines[k]->pool_idle_bytes;
    }
    if (buffers) *buffers = nb;
    if (bytes) *bytes = by;
}

}  // namespace engine
}  // namespace oval
