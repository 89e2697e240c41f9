// oval_capi.cpp — implementation of the C ABI declared in include/oval_hip.h.
//
// There is no CPU implementation behind this ABI: every fill and validate goes
// through the gfx950 kernels, and a process without a usable device gets
// OVAL_E_NODEVICE / OVAL_E_HIP, never a silent fallback.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/oval_hip.h"
#include "oval_capi_internal.h"
#include "oval_kernels.h"
#include "oval_gostr.h"
#include "oval_trimspace.h"

static_assert(sizeof(oval_obj_desc) == 64, "descriptor must be 64 bytes");
static_assert(offsetof(oval_obj_desc, key) == 16, "layout");
static_assert(offsetof(oval_obj_desc, write_count) == 28, "layout");
static_assert(offsetof(oval_obj_desc, first_unit) == 32, "layout");
static_assert(offsetof(oval_obj_desc, worker_id) == 36, "layout");
static_assert(offsetof(oval_obj_desc, unix_micro) == 40, "layout");
static_assert(offsetof(oval_obj_desc, body_offset) == 48, "layout");
static_assert(offsetof(oval_obj_desc, n_units) == 56, "layout");
static_assert(offsetof(oval_obj_desc, bucket_len) == 60, "layout");
static_assert(offsetof(oval_obj_desc, flags) == 62, "layout");
static_assert(sizeof(oval_obj_result) == 16, "result must be 16 bytes");

struct oval_ctx {
    int         device      = 0;
    hipStream_t stream      = nullptr;
    bool        own_stream  = false;
    hipStream_t copy_stream = nullptr;   // second stream for overlapped host copies
    hipStream_t copy_stream2 = nullptr;  // odd slices of the host-memory entry points copy here: two copy engines at work
    int         fill_threads = 256;    // workgroup sizes of the two kernels
    int         validate_threads = 256;
    unsigned long long* tallies_dev = nullptr;   // striped bad-unit tallies of the validate kernels, zero between passes
    // staging ring of the host-memory entry points, created on their first use
    static constexpr int kRingSlots = 4;
    void*       ring_dev = nullptr;                // kRingSlots slices
    hipEvent_t  ring_ready[kRingSlots] = {};       // slot holds what its consumer waits for
    hipEvent_t  ring_free[kRingSlots] = {};        // slot's consumer is done with it
    // what else the host-memory entry points need, owned by the context and grow-only
    oval_layout* host_layout = nullptr;
    oval_obj_result* host_res_dev = nullptr;
    uint64_t    host_res_cap = 0;
    unsigned long long* host_cnt_dev = nullptr;
    void*       bounce_host = nullptr;             // kRingSlots pinned slices for pageable bodies
    hipEvent_t  bounce_ev[kRingSlots] = {};        // the copy engine is done with the bounce slot
};

struct oval_layout {
    int       device = 0;
    void*     descs_dev = nullptr;
    uint32_t* chunk_first_dev = nullptr;
    void*     chunk_desc_dev = nullptr;   // per-chunk descriptor copies, only while `one_level`
    bool      one_level = false;
    bool      small_fill = false, small_validate = false;   // objects of a few units: the kernels that resolve a chunk's objects at once
    bool      big_validate = false;     // objects of megabytes: one lookup per workgroup of 1024 threads
    uint64_t  n_obj = 0;
    uint64_t  total_units = 0;
    uint64_t  span_bytes = 0;
    uint32_t  n_tiles = 0;
    bool      no_fill = false;   // some descriptor carries OVAL_DESC_NO_FILL
    uint64_t  cap_obj = 0;       // descriptors the device table can hold
    uint32_t  cap_tiles = 0;    // tiles the chunk table has room for (kChunksPerTile entries each)
    uint32_t  cap_desc_tiles = 0;   // likewise for chunk_desc_dev
    void*     stage_host = nullptr; // pinned copy of the caller's descriptors (or records) on their way to the device
    uint64_t  cap_stage_bytes = 0;
    void*     compact_dev = nullptr; // records, segments and block sums of a layout built from compact records
    uint64_t  cap_compact_bytes = 0;
    hipStream_t built_on = nullptr; // stream the tables were last built on ...
    hipEvent_t  built_ev = nullptr; // ... and the event that says they are complete
    // A layout that has been rebuilt ahead of time (oval_layout_reset_ahead) has
    // every later pass recorded here, so that its next rebuild, on another stream,
    // waits for the passes that still read its tables and for nothing else.
    bool        track_use = false;
    mutable hipEvent_t used_ev = nullptr;
    mutable bool used = false;      // used_ev has been recorded since the last rebuild
};

struct oval_timer {
    hipEvent_t start = nullptr, stop = nullptr;
};

namespace {

// The kernels index the descriptor table in 16-byte quarters with 32-bit arithmetic (4 * object + quarter).
constexpr uint64_t kMaxObjects = 1ull << 30;

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

int oval_internal_fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
int oval_internal_caught(void)
{
    try {
        throw;
    } catch (const std::bad_alloc&) {
        return oval_internal_fail(OVAL_E_NOMEM, "out of host memory inside the library");
    } catch (const std::exception& e) {
        return oval_internal_fail(OVAL_E_INTERNAL, "unexpected failure inside the library: %s", e.what());
    } catch (...) {
        return oval_internal_fail(OVAL_E_INTERNAL, "unexpected failure inside the library");
    }
}
int oval_internal_caught_text(char* err, size_t err_cap)
{
    const int rc = oval_internal_caught();
    if (err && err_cap) {
        std::strncpy(err, g_err.c_str(), err_cap - 1);
        err[err_cap - 1] = '\0';
    }
    return rc;
}
void oval_internal_parallel_copy(void* dst, const void* src, size_t n);   // memcpy, on several threads when large
int oval_internal_ctx_device(const oval_ctx* ctx) { return ctx->device; }
hipStream_t oval_internal_ctx_stream(const oval_ctx* ctx) { return ctx->stream; }

// ---- waits on the device that end ------------------------------------------------
// hipEventSynchronize / hipStreamSynchronize never return while the device keeps
// answering "not ready": a kernel that never finishes would take every host thread
// that waits for it along.  The host-memory entry points and the pipeline behind
// the pattern API wait through these two instead: they poll, back off, and give up
// after the process-wide limit (oval_set_device_timeout_ms, 30 s by default) with
// hipErrorNotReady and the thread's timed-out mark set.
namespace {
std::atomic<uint64_t> g_device_timeout_ms{30000};
std::atomic<uint64_t> g_small_fill_bytes{OVAL_SMALL_FILL_BYTES_DEFAULT};          // oval_set_small_object_bytes
std::atomic<uint64_t> g_small_validate_bytes{OVAL_SMALL_VALIDATE_BYTES_DEFAULT};
std::atomic<uint64_t> g_big_validate_bytes{OVAL_BIG_VALIDATE_BYTES_DEFAULT};      // oval_set_big_object_bytes
thread_local bool g_wait_timed_out = false;

uint64_t mono_ns()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

template <class Query>
hipError_t bounded_wait(const Query& query, bool sleepy)
{
    uint64_t t0 = 0;
    for (uint32_t polls = 0;; ++polls) {
        const hipError_t q = query();
        if (q != hipErrorNotReady) return q;
        (void)hipGetLastError();                                 // "not ready" is not an error to keep
        if (!sleepy && (polls & 63u) != 63u) {
            __builtin_ia32_pause();
            continue;
        }
        const uint64_t now = mono_ns();
        if (!t0) t0 = now;
        const uint64_t waited = now - t0, limit = g_device_timeout_ms.load(std::memory_order_relaxed) * 1000000ull;
        if (limit && waited > limit) {
            g_wait_timed_out = true;
            return hipErrorNotReady;
        }
        // short waits spin; a wait that has lasted milliseconds gives the core away between looks.  A sleepy
        // waiter naps from the start, briefly (its thread's timer slack decides how briefly: the pipeline's
        // lane threads set theirs to a microsecond).
        if (sleepy || waited > 50000000ull) {
            const timespec nap = {0, sleepy && waited < 2000000ull ? 15000 : 200000};
            nanosleep(&nap, nullptr);
        } else if (waited > 2000000ull) {
            sched_yield();
        }
    }
}
}  // namespace

hipError_t oval_internal_wait_event(hipEvent_t ev, bool sleepy)
{
    return bounded_wait([ev] { return hipEventQuery(ev); }, sleepy);
}

hipError_t oval_internal_wait_stream(hipStream_t stream)
{
    return bounded_wait([stream] { return hipStreamQuery(stream); }, false);
}

bool oval_internal_wait_timed_out(bool clear)
{
    const bool t = g_wait_timed_out;
    if (clear) g_wait_timed_out = false;
    return t;
}

uint64_t oval_internal_device_timeout_ns(void) { return g_device_timeout_ms.load(std::memory_order_relaxed) * 1000000ull; }

const char* oval_internal_wait_error(hipError_t e)
{
    return e == hipErrorNotReady ? "the device did not finish within the time limit (oval_set_device_timeout_ms)"
                                 : hipGetErrorString(e);
}

namespace {

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(e_ == hipErrorOutOfMemory ? OVAL_E_NOMEM : OVAL_E_HIP,           \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                  \
    } while (0)

oval::LayoutView view_of(const oval_layout* l)
{
    oval::LayoutView v;
    v.descs = l->descs_dev;
    v.chunk_first = l->chunk_first_dev;
    v.chunk_desc = l->one_level ? l->chunk_desc_dev : nullptr;
    v.small_fill = l->small_fill;
    v.small_validate = l->small_validate;
    v.big_validate = l->big_validate;
    v.n_obj = (uint32_t)l->n_obj;
    v.n_tiles = l->n_tiles;
    v.span_units = (uint32_t)(l->span_bytes / OVAL_UNIT_SIZE);
    v.total_units = l->total_units;
    return v;
}

// Launch shape: workgroups of 256 threads measured best for both kernels
// (128 / 512 / 1024: fill -15 / -1 / -10 %, validate -2 / 0 / -7 %);
// oval_ctx_set_tuning() overrides it for such sweeps.
oval::LaunchCfg cfg_of(const oval_ctx* c)
{
    oval::LaunchCfg k;
    k.fill_threads = c->fill_threads;
    k.validate_threads = c->validate_threads;
    return k;
}

uint32_t tiles_of(const oval_layout* l)
{
    return l->n_tiles;
}

// true when strings.TrimSpace(name) != name, i.e. the expected name itself
// starts or ends with white space
bool go_space_at_edges(const unsigned char* s, size_t n)
{
    int a, b;
    oval::trim_space([&](int i) { return (uint32_t)s[i]; }, (int)n, a, b);
    return !(a == 0 && b == (int)n);
}
}  // namespace

extern "C" {

const char* oval_last_error(void) { return g_err.c_str(); }

const char* oval_version(void) { return "oval_hip 0.1 (gfx950)"; }

int oval_set_device_timeout_ms(uint64_t ms)
try {
    g_device_timeout_ms.store(ms, std::memory_order_relaxed);
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_set_small_object_bytes(uint64_t fill_bytes, uint64_t validate_bytes)
try {
    g_small_fill_bytes.store(fill_bytes, std::memory_order_relaxed);
    g_small_validate_bytes.store(validate_bytes, std::memory_order_relaxed);
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_set_big_object_bytes(uint64_t validate_bytes)
try {
    g_big_validate_bytes.store(validate_bytes, std::memory_order_relaxed);
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_device_count(int* n_out)
try {
    if (!n_out) return fail(OVAL_E_INVALID, "n_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *n_out = 0; return fail(OVAL_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n_out = n;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_ctx_create(int device_id, void* stream, oval_ctx** out)
try {
    if (!out) return fail(OVAL_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(OVAL_E_NODEVICE, "no HIP device is visible; the pattern engine has no CPU path");
    if (device_id < 0 || device_id >= n) return fail(OVAL_E_INVALID, "device %d out of range [0,%d)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(OVAL_E_NODEVICE, "device %d is %s; this library carries gfx950 code only", device_id,
                    prop.gcnArchName);
    oval_ctx* c = new (std::nothrow) oval_ctx;
    if (!c) return fail(OVAL_E_NOMEM, "out of host memory");
    c->device = device_id;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(OVAL_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->own_stream = true;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(OVAL_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&c->copy_stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&c->tallies_dev, oval::kTallyBytes);
    if (e == hipSuccess) e = hipMemset(c->tallies_dev, 0, oval::kTallyBytes);
    if (e != hipSuccess) {
        if (c->tallies_dev) (void)hipFree(c->tallies_dev);
        if (c->copy_stream2) (void)hipStreamDestroy(c->copy_stream2);
        (void)hipStreamDestroy(c->copy_stream);
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(OVAL_E_NOMEM, "hipMalloc(tallies): %s", hipGetErrorString(e));
    }
    // where this GPU keeps the static scratch of oval_validate_noalloc
    e = oval::noalloc_scratch_init(device_id);
    if (e != hipSuccess) {
        (void)hipFree(c->tallies_dev);
        (void)hipStreamDestroy(c->copy_stream2);
        (void)hipStreamDestroy(c->copy_stream);
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(OVAL_E_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device_id, hipGetErrorString(e));
    }
    *out = c;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_ctx_destroy(oval_ctx* ctx)
try {
    if (!ctx) return OVAL_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->copy_stream2);
    if (ctx->host_layout) (void)oval_layout_destroy(ctx, ctx->host_layout);
    (void)hipStreamDestroy(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream2);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->tallies_dev) (void)hipFree(ctx->tallies_dev);
    if (ctx->ring_dev) (void)hipFree(ctx->ring_dev);
    if (ctx->host_res_dev) (void)hipFree(ctx->host_res_dev);
    if (ctx->host_cnt_dev) (void)hipFree(ctx->host_cnt_dev);
    if (ctx->bounce_host) (void)hipHostFree(ctx->bounce_host);
    for (int i = 0; i < oval_ctx::kRingSlots; ++i) {
        if (ctx->ring_ready[i]) (void)hipEventDestroy(ctx->ring_ready[i]);
        if (ctx->ring_free[i]) (void)hipEventDestroy(ctx->ring_free[i]);
        if (ctx->bounce_ev[i]) (void)hipEventDestroy(ctx->bounce_ev[i]);
    }
    delete ctx;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_ctx_sync(oval_ctx* ctx)
try {
    if (!ctx) return fail(OVAL_E_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

}  // extern "C"

// Waits for the context's main stream AND its two copy streams: the host-memory entry points leave slice
// copies on the latter, which oval_ctx_sync does not cover.
int oval_internal_ctx_sync_all(oval_ctx* ctx)
{
    if (!ctx) return fail(OVAL_E_INVALID, "ctx is NULL");
    const hipError_t e1 = oval_internal_wait_stream(ctx->stream);
    const hipError_t e2 = oval_internal_wait_stream(ctx->copy_stream);
    const hipError_t e3 = oval_internal_wait_stream(ctx->copy_stream2);
    const hipError_t e = e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3;
    if (e != hipSuccess) return fail(OVAL_E_HIP, "waiting for the context's streams: %s", oval_internal_wait_error(e));
    return OVAL_OK;
}

extern "C" {

int oval_ctx_stream(oval_ctx* ctx, void** stream_out)
try {
    if (!ctx || !stream_out) return fail(OVAL_E_INVALID, "NULL argument");
    *stream_out = (void*)ctx->stream;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_ctx_set_tuning(oval_ctx* ctx, int fill_threads, int validate_threads)
try {
    if (!ctx) return fail(OVAL_E_INVALID, "ctx is NULL");
    auto ok = [](int t) { return t <= 0 || t == 128 || t == 256 || t == 512 || t == 1024; };
    if (!ok(fill_threads) || !ok(validate_threads))
        return fail(OVAL_E_INVALID, "workgroup sizes are 128, 256, 512 or 1024 threads (0 = default)");
    ctx->fill_threads = fill_threads > 0 ? fill_threads : 256;
    ctx->validate_threads = validate_threads > 0 ? validate_threads : 256;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_dev_alloc(oval_ctx* ctx, size_t bytes, void** dptr_out)
try {
    if (!ctx || !dptr_out) return fail(OVAL_E_INVALID, "NULL argument");
    *dptr_out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(dptr_out, bytes ? bytes : 1));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_dev_free(oval_ctx* ctx, void* dptr)
try {
    if (!ctx) return fail(OVAL_E_INVALID, "ctx is NULL");
    if (!dptr) return OVAL_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipFree(dptr));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_pinned_alloc(oval_ctx* ctx, size_t bytes, void** hptr_out)
try {
    if (!ctx || !hptr_out) return fail(OVAL_E_INVALID, "NULL argument");
    *hptr_out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipHostMalloc(hptr_out, bytes ? bytes : 1, hipHostMallocDefault));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_pinned_free(oval_ctx* ctx, void* hptr)
try {
    if (!ctx) return fail(OVAL_E_INVALID, "ctx is NULL");
    if (!hptr) return OVAL_OK;
    HIP_TRY(hipHostFree(hptr));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_copy_h2d(oval_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes)
try {
    if (!ctx || (!dst_dev && bytes) || (!src_host && bytes)) return fail(OVAL_E_INVALID, "NULL argument");
    if (!bytes) return OVAL_OK;
    HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_copy_d2h(oval_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes)
try {
    if (!ctx || (!dst_host && bytes) || (!src_dev && bytes)) return fail(OVAL_E_INVALID, "NULL argument");
    if (!bytes) return OVAL_OK;
    HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_desc_init(oval_obj_desc* d, const char* bucket, size_t bucket_len, const char* key,
                   size_t key_len, int64_t write_count, int64_t worker_id, uint64_t unix_micro,
                   uint64_t body_offset, uint32_t first_unit, uint32_t n_units)
try {
    if (!d || (!bucket && bucket_len) || (!key && key_len)) return fail(OVAL_E_INVALID, "NULL argument");
    const unsigned char* b = (const unsigned char*)bucket;
    const unsigned char* k = (const unsigned char*)key;
    if (bucket_len > OVAL_MAX_BUCKET_LEN) bucket_len = OVAL_MAX_BUCKET_LEN;   // pattern.go:74-76,119-121
    if (key_len > OVAL_MAX_KEY_LEN)
        return fail(OVAL_E_FIELD, "key is %zu bytes; \"%%-12s\" does not truncate, the unit would not be 256 bytes", key_len);
    // fmt "%-16s%-12s" (pattern.go:95,98) pads to a RUNE count
    const size_t bfield = oval::go_padded_len(b, bucket_len, OVAL_MAX_BUCKET_LEN);
    const size_t kfield = oval::go_padded_len(k, key_len, OVAL_MAX_KEY_LEN);
    if (body_offset % OVAL_UNIT_SIZE) return fail(OVAL_E_INVALID, "body_offset %llu is not a multiple of 256",
                                                  (unsigned long long)body_offset);
    std::memset(d, 0, sizeof *d);
    std::memset(d->bucket, ' ', sizeof d->bucket);
    std::memset(d->key, ' ', sizeof d->key);
    std::memcpy(d->bucket, b, bucket_len);
    std::memcpy(d->key, k, key_len);
    d->write_count = (uint32_t)write_count;   // pattern.go:101,154
    d->first_unit = first_unit;
    d->worker_id = (uint32_t)worker_id;       // pattern.go:105
    d->unix_micro = unix_micro;
    d->body_offset = body_offset;
    d->n_units = n_units;
    d->bucket_len = (uint8_t)bucket_len;
    d->key_len = (uint8_t)key_len;
    uint16_t flags = 0;
    if (go_space_at_edges(b, bucket_len)) flags |= OVAL_DESC_BUCKET_SLOW;
    if (go_space_at_edges(k, key_len)) flags |= OVAL_DESC_KEY_SLOW;
    // validDataUnit compares the worker ID as a Go int (pattern.go:166-167): a
    // value outside uint32 can never equal what the unit holds
    if (worker_id < 0 || worker_id > 0xFFFFFFFFll) flags |= OVAL_DESC_WORKER_OOR;
    // names with multi-byte runes pad past their field in the reference, whose
    // Generate then fails (pattern.go:87-89): such an object can be validated
    // but must not be filled
    if (bfield != OVAL_MAX_BUCKET_LEN || kfield != OVAL_MAX_KEY_LEN) flags |= OVAL_DESC_NO_FILL;
    d->flags = flags;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

// A few helper threads for host work that is too large for one core (staging
// copies of pageable bodies, scanning and copying a million descriptors): tasks
// 0 .. n-1 are handed out through a counter, the caller takes part, one run at
// a time.  The threads are started on first use and sleep until the process ends.
namespace {
constexpr int kPoolHelpers = 7;          // + the calling thread

class TaskPool {
public:
    static TaskPool& get()
    {
        static TaskPool* p = new TaskPool;   // never destroyed: its threads sleep until the process ends
        return *p;
    }

    void run(size_t n, const std::function<void(size_t)>& fn)
    {
        if (n == 0) return;
        if (n == 1) { fn(0); return; }
        std::lock_guard<std::mutex> call(call_mu_);
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (threads_ == 0)
                for (; threads_ < kPoolHelpers; ++threads_) std::thread([this] { worker(); }).detach();
            fn_ = &fn;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            pending_ = threads_;
            ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    void work()
    {
        for (;;) {
            const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_) return;
            (*fn_)(i);
        }
    }

    void worker()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            lk.unlock();
            work();
            lk.lock();
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }

    std::mutex call_mu_;                 // one run at a time
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    int threads_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0};
};
}  // namespace

// true when the name certainly has no white space at either edge: both edge
// bytes are ASCII and not space (a multi-byte white-space rune starts with a
// byte >= 0xC2 and ends with one >= 0x80)
static inline bool edges_plain(const uint8_t* s, size_t n)
{
    return n == 0 || (s[0] < 0x80u && !oval::ascii_space(s[0]) && s[n - 1] < 0x80u && !oval::ascii_space(s[n - 1]));
}

// The OVAL_DESC_*_SLOW bits a descriptor must carry: they follow from the
// names, so they are derived here rather than trusted (a table assembled without
// oval_desc_init would otherwise validate "name " clean on the 128-bit path
// where the reference's TrimSpace compare reports every unit, pattern.go:140,147).
static inline uint16_t slow_flags_of(const oval_obj_desc& d)
{
    uint16_t f = 0;
    if (!edges_plain(d.bucket, d.bucket_len) && go_space_at_edges(d.bucket, d.bucket_len)) f |= OVAL_DESC_BUCKET_SLOW;
    if (!edges_plain(d.key, d.key_len) && go_space_at_edges(d.key, d.key_len)) f |= OVAL_DESC_KEY_SLOW;
    return f;
}

// Checks descriptors [begin, end) of a set and computes their totals.  `fix`
// receives the indices of the descriptors whose OVAL_DESC_*_SLOW bits differ
// from what their names demand.
struct ScanPart {
    uint64_t begin = 0, end = 0;
    uint64_t total_units = 0, last_end = 0;
    bool no_fill = false;
    std::vector<uint64_t> fix;
    int rc = OVAL_OK;
};

static int scan_range(const oval_obj_desc* descs, ScanPart* part)
{
    bool no_fill = false;
    // "the previous object" of the range's first one is the element before it
    uint64_t prev_end = part->begin ? descs[part->begin - 1].body_offset +
                                          (uint64_t)descs[part->begin - 1].n_units * OVAL_UNIT_SIZE : 0;
    uint64_t total_units = 0;
    const uint16_t slow_bits = OVAL_DESC_BUCKET_SLOW | OVAL_DESC_KEY_SLOW;
    part->fix.clear();
    for (uint64_t i = part->begin; i < part->end; ++i) {
        const oval_obj_desc& d = descs[i];
        if (d.body_offset % OVAL_UNIT_SIZE)
            return fail(OVAL_E_LAYOUT, "object %llu: body_offset %llu is not a multiple of 256",
                        (unsigned long long)i, (unsigned long long)d.body_offset);
        if (d.body_offset < prev_end)
            return fail(OVAL_E_LAYOUT, "object %llu: body at %llu overlaps or precedes the previous object (ends %llu)",
                        (unsigned long long)i, (unsigned long long)d.body_offset, (unsigned long long)prev_end);
        if (d.bucket_len > OVAL_MAX_BUCKET_LEN || d.key_len > OVAL_MAX_KEY_LEN)
            return fail(OVAL_E_LAYOUT, "object %llu: name lengths %u/%u exceed 16/12", (unsigned long long)i,
                        d.bucket_len, d.key_len);
        prev_end = d.body_offset + (uint64_t)d.n_units * OVAL_UNIT_SIZE;
        total_units += d.n_units;
        no_fill |= (d.flags & OVAL_DESC_NO_FILL) != 0;
        if ((d.flags & slow_bits) != slow_flags_of(d)) part->fix.push_back(i);
        if (prev_end / OVAL_UNIT_SIZE >= 0xFFFFFF00ull)
            return fail(OVAL_E_LAYOUT, "batch spans %llu bytes; the limit is just under 1 TiB",
                        (unsigned long long)prev_end);
    }
    part->no_fill = no_fill;
    part->total_units = total_units;
    part->last_end = prev_end;
    return OVAL_OK;
}

// The whole set; large ones in up to eight ranges side by side (a range only
// needs the element before its first one).  The first failing range, in table
// order, is scanned again on the calling thread, whose error text the caller reads.
static int scan_descs(const oval_obj_desc* descs, uint64_t n_obj, uint64_t* total_units_out,
                      uint64_t* span_bytes_out, bool* no_fill_out, std::vector<uint64_t>* fix)
{
    if (n_obj > kMaxObjects) return fail(OVAL_E_LAYOUT, "too many objects (%llu): a batch holds at most 2^30", (unsigned long long)n_obj);
    const uint64_t n_parts = n_obj >= (128u << 10) ? (n_obj >> 16 < 8 ? n_obj >> 16 : 8) : 1;
    std::vector<ScanPart> parts(n_parts);
    for (uint64_t p = 0; p < n_parts; ++p) {
        parts[p].begin = n_obj * p / n_parts;
        parts[p].end = n_obj * (p + 1) / n_parts;
    }
    TaskPool::get().run(n_parts, [&](size_t p) { parts[p].rc = scan_range(descs, &parts[p]); });
    uint64_t total_units = 0;
    bool no_fill = false;
    fix->clear();
    for (ScanPart& part : parts) {
        if (part.rc) return n_parts == 1 ? part.rc : scan_range(descs, &part);
        total_units += part.total_units;
        no_fill |= part.no_fill;
        fix->insert(fix->end(), part.fix.begin(), part.fix.end());
    }
    *no_fill_out = no_fill;
    *total_units_out = total_units;
    *span_bytes_out = n_obj ? parts.back().last_end : 0;
    return OVAL_OK;
}

// What a rebuild of a layout is to hold, known before anything touches the device.
struct LayoutShape {
    uint64_t n_obj = 0, total_units = 0, span_bytes = 0;
    uint64_t stage_bytes = 0;     // pinned staging the upload needs
    uint64_t compact_bytes = 0;   // device memory for records, segments and block sums (compact form only)
    bool no_fill = false;
    uint32_t n_tiles = 0;
    bool one_level = false;
    bool small_fill = false, small_validate = false;
    bool big_validate = false;
};

// First half of a rebuild: capacity (grow-only) and ordering against whatever
// still reads or writes the old tables.  `ahead`: the tables go to the device
// and are built on the context's copy stream, behind the layout's own last pass
// only, instead of behind everything queued on the context's stream.
static int layout_prepare(oval_ctx* ctx, oval_layout* l, LayoutShape& sh, bool ahead, hipStream_t* build_stream_out)
{
    const uint64_t span_units = sh.span_bytes / OVAL_UNIT_SIZE;
    sh.n_tiles = (uint32_t)((span_units + oval::kUnitsPerTile - 1) / oval::kUnitsPerTile);
    // Objects that average under the small-object limits (gaps included) take the kernels that resolve all objects
    // of a chunk with one vector load instead of walking them one dependent scalar load at a time; such a fill
    // needs no per-chunk descriptor copies.
    const uint64_t avg_bytes = sh.n_obj ? sh.span_bytes / sh.n_obj : 0;
    sh.small_fill = sh.n_obj != 0 && avg_bytes < g_small_fill_bytes.load(std::memory_order_relaxed);
    sh.small_validate = sh.n_obj != 0 && avg_bytes < g_small_validate_bytes.load(std::memory_order_relaxed);
    sh.one_level = !sh.small_fill && sh.n_obj != 0 && avg_bytes < (256ull << 10);
    sh.big_validate = sh.n_obj != 0 && avg_bytes >= g_big_validate_bytes.load(std::memory_order_relaxed);
    const uint64_t n_obj = sh.n_obj;
    const uint32_t n_tiles = sh.n_tiles;
    const bool one_level = sh.one_level;
    HIP_TRY(hipSetDevice(ctx->device));
    // Until every table is in place the layout describes nothing, so that a
    // failure below cannot leave counts that point into freed or missing tables.
    l->n_obj = 0; l->n_tiles = 0; l->total_units = 0; l->span_bytes = 0; l->one_level = false; l->small_fill = false; l->small_validate = false; l->big_validate = false; l->no_fill = false;
    const bool grow = n_obj > l->cap_obj || !l->descs_dev || n_tiles > l->cap_tiles || !l->chunk_first_dev ||
                      (one_level && (n_tiles > l->cap_desc_tiles || !l->chunk_desc_dev)) ||
                      sh.stage_bytes > l->cap_stage_bytes || sh.compact_bytes > l->cap_compact_bytes;
    hipStream_t const build_stream = ahead ? ctx->copy_stream : ctx->stream;
    *build_stream_out = build_stream;
    if (grow) {
        // a pass that still uses the old tables must finish before they are freed
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (l->built_ev) HIP_TRY(hipEventSynchronize(l->built_ev));
    } else if (l->built_ev) {
        // Same capacity: the new tables are written in stream order behind the
        // passes already issued on this stream, so the device is not waited for;
        // only the pinned staging copy must have left for the device, which the
        // previous build's event covers (long complete in the usual case).
        HIP_TRY(hipEventSynchronize(l->built_ev));
    }
    if (n_obj > l->cap_obj || !l->descs_dev) {
        if (l->descs_dev) { (void)hipFree(l->descs_dev); l->descs_dev = nullptr; l->cap_obj = 0; }
        const uint64_t cap = n_obj ? n_obj : 1;
        HIP_TRY(hipMalloc(&l->descs_dev, cap * sizeof(oval_obj_desc)));
        l->cap_obj = cap;
    }
    if (n_tiles > l->cap_tiles || !l->chunk_first_dev) {
        if (l->chunk_first_dev) { (void)hipFree(l->chunk_first_dev); l->chunk_first_dev = nullptr; l->cap_tiles = 0; }
        const uint32_t cap = n_tiles ? n_tiles : 1;
        HIP_TRY(hipMalloc((void**)&l->chunk_first_dev, (size_t)cap * oval::kChunksPerTile * 4));
        l->cap_tiles = cap;
    }
    // Objects that average under 256 KiB get a 64-byte copy of its first object's
    // descriptor per chunk (1.6 % of the body size): the fill then starts with one
    // lookup instead of two dependent ones, worth 6-10 % on 4 KiB ... 64 KiB
    // objects; for larger objects the extra table reads cost more than they save.
    if (one_level && (n_tiles > l->cap_desc_tiles || !l->chunk_desc_dev)) {
        if (l->chunk_desc_dev) { (void)hipFree(l->chunk_desc_dev); l->chunk_desc_dev = nullptr; l->cap_desc_tiles = 0; }
        const uint32_t cap = n_tiles ? n_tiles : 1;
        HIP_TRY(hipMalloc(&l->chunk_desc_dev, (size_t)cap * oval::kChunksPerTile * sizeof(oval_obj_desc)));
        l->cap_desc_tiles = cap;
    }
    if (sh.stage_bytes > l->cap_stage_bytes) {
        if (l->stage_host) { (void)hipHostFree(l->stage_host); l->stage_host = nullptr; l->cap_stage_bytes = 0; }
        HIP_TRY(hipHostMalloc(&l->stage_host, sh.stage_bytes, hipHostMallocDefault));
        l->cap_stage_bytes = sh.stage_bytes;
    }
    if (sh.compact_bytes > l->cap_compact_bytes) {
        if (l->compact_dev) { (void)hipFree(l->compact_dev); l->compact_dev = nullptr; l->cap_compact_bytes = 0; }
        HIP_TRY(hipMalloc(&l->compact_dev, sh.compact_bytes));
        l->cap_compact_bytes = sh.compact_bytes;
    }
    if (!grow) {
        // a build still running on another stream writes the same tables
        if (l->built_ev && l->built_on != build_stream) HIP_TRY(hipStreamWaitEvent(build_stream, l->built_ev, 0));
        if (ahead) {
            // the passes that read the old tables were issued on the context's stream
            if (!l->used_ev) HIP_TRY(hipEventCreateWithFlags(&l->used_ev, hipEventDisableTiming));
            if (!l->track_use) {
                // not recorded so far: everything queued on that stream counts as a reader, this once
                HIP_TRY(hipEventRecord(l->used_ev, ctx->stream));
                l->used = true;
            }
            if (l->used) HIP_TRY(hipStreamWaitEvent(build_stream, l->used_ev, 0));
        }
    }
    if (ahead) {
        if (!l->used_ev) HIP_TRY(hipEventCreateWithFlags(&l->used_ev, hipEventDisableTiming));
        l->track_use = true;
    }
    l->used = false;
    return OVAL_OK;
}

// Second half: the chunk index over the descriptor table now on its way on
// `build_stream`, the event that says the tables are complete, and the counts.
static int layout_finish(oval_layout* l, const LayoutShape& sh, hipStream_t build_stream)
{
    HIP_TRY(oval::launch_build_chunk_table(build_stream, l->descs_dev, (uint32_t)sh.n_obj, l->chunk_first_dev,
                                           sh.one_level ? l->chunk_desc_dev : nullptr, sh.n_tiles));
    if (!l->built_ev) HIP_TRY(hipEventCreateWithFlags(&l->built_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(l->built_ev, build_stream));
    l->built_on = build_stream;
    l->one_level = sh.one_level;
    l->small_fill = sh.small_fill;
    l->small_validate = sh.small_validate;
    l->big_validate = sh.big_validate;
    l->n_obj = sh.n_obj;
    l->no_fill = sh.no_fill;
    l->total_units = sh.total_units;
    l->span_bytes = sh.span_bytes;
    l->n_tiles = sh.n_tiles;
    return OVAL_OK;
}

static int layout_rebuild(oval_ctx* ctx, oval_layout* l, const oval_obj_desc* descs, uint64_t n_obj, bool ahead)
{
    if (!ctx || !l || (!descs && n_obj)) return fail(OVAL_E_INVALID, "NULL argument");
    if (l->device != ctx->device) return fail(OVAL_E_INVALID, "layout belongs to device %d, context to %d", l->device, ctx->device);
    LayoutShape sh;
    std::vector<uint64_t> fix;
    int rc = scan_descs(descs, n_obj, &sh.total_units, &sh.span_bytes, &sh.no_fill, &fix);
    if (rc) return rc;
    sh.n_obj = n_obj;
    sh.stage_bytes = n_obj * sizeof(oval_obj_desc);
    hipStream_t build_stream = nullptr;
    if ((rc = layout_prepare(ctx, l, sh, ahead, &build_stream))) return rc;
    // The caller's array is not retained and may be pageable: it is copied to a
    // pinned buffer of the layout's here, and goes to the device from there in
    // stream order, so this call does not wait for the device.
    if (n_obj) {
        oval_internal_parallel_copy(l->stage_host, descs, n_obj * sizeof(oval_obj_desc));
        oval_obj_desc* staged = static_cast<oval_obj_desc*>(l->stage_host);
        const uint16_t slow_bits = OVAL_DESC_BUCKET_SLOW | OVAL_DESC_KEY_SLOW;
        for (uint64_t i : fix) staged[i].flags = (uint16_t)((staged[i].flags & ~slow_bits) | slow_flags_of(staged[i]));
        HIP_TRY(hipMemcpyAsync(l->descs_dev, l->stage_host, n_obj * sizeof(oval_obj_desc), hipMemcpyHostToDevice,
                               build_stream));
    }
    return layout_finish(l, sh, build_stream);
}

int oval_layout_reset(oval_ctx* ctx, oval_layout* l, const oval_obj_desc* descs, uint64_t n_obj)
try {
    return layout_rebuild(ctx, l, descs, n_obj, false);
} catch (...) {
    return oval_internal_caught();
}

int oval_layout_reset_ahead(oval_ctx* ctx, oval_layout* l, const oval_obj_desc* descs, uint64_t n_obj)
try {
    return layout_rebuild(ctx, l, descs, n_obj, true);
} catch (...) {
    return oval_internal_caught();
}

static_assert(sizeof(oval_obj_rec) == 12, "record must be 12 bytes");
static_assert(sizeof(oval_obj_segment) == 48 && offsetof(oval_obj_segment, key_id_base) == 16 &&
              offsetof(oval_obj_segment, worker_id) == 32 && offsetof(oval_obj_segment, flags) == 46, "segment layout");

int oval_segment_init(oval_obj_segment* seg, const char* bucket, size_t bucket_len, int64_t worker_id,
                      uint64_t unix_micro, uint64_t key_id_base, uint32_t first_obj, uint32_t n_obj)
try {
    if (!seg) return fail(OVAL_E_INVALID, "NULL argument");
    // the same formatting and flags as a descriptor's, taken from one built for a key of the reference's shape
    oval_obj_desc d;
    int rc = oval_desc_init(&d, bucket, bucket_len, "ov0000000000", 12, 0, worker_id, unix_micro, 0, 0, 0);
    if (rc) return rc;
    std::memset(seg, 0, sizeof *seg);
    std::memcpy(seg->bucket, d.bucket, sizeof seg->bucket);
    seg->key_id_base = key_id_base;
    seg->unix_micro = unix_micro;
    seg->worker_id = d.worker_id;
    seg->first_obj = first_obj;
    seg->n_obj = n_obj;
    seg->bucket_len = d.bucket_len;
    seg->flags = d.flags;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_layout_reset_compact(oval_ctx* ctx, oval_layout* l, const oval_obj_segment* segs, uint64_t n_segs,
                              const oval_obj_rec* recs, uint64_t n_obj, uint32_t flags)
try {
    if (!ctx || !l || (!recs && n_obj) || (!segs && n_segs)) return fail(OVAL_E_INVALID, "NULL argument");
    if (flags & ~OVAL_RESET_AHEAD) return fail(OVAL_E_INVALID, "unknown flags 0x%x", flags);
    if (l->device != ctx->device) return fail(OVAL_E_INVALID, "layout belongs to device %d, context to %d", l->device, ctx->device);
    if (n_obj > kMaxObjects) return fail(OVAL_E_LAYOUT, "too many objects (%llu): a batch holds at most 2^30", (unsigned long long)n_obj);
    if (n_segs > n_obj) return fail(OVAL_E_LAYOUT, "%llu segments for %llu objects: a segment holds at least one",
                                    (unsigned long long)n_segs, (unsigned long long)n_obj);
    LayoutShape sh;
    sh.n_obj = n_obj;
    // the segments cover the records in order; per segment the units and the highest object ID
    uint64_t next = 0;
    for (uint64_t k = 0; k < n_segs; ++k) {
        if (segs[k].first_obj != next || segs[k].n_obj == 0)
            return fail(OVAL_E_LAYOUT, "segment %llu starts at record %u with %u objects; record %llu is the next uncovered one",
                        (unsigned long long)k, segs[k].first_obj, segs[k].n_obj, (unsigned long long)next);
        if (segs[k].bucket_len > OVAL_MAX_BUCKET_LEN)
            return fail(OVAL_E_LAYOUT, "segment %llu: bucket length %u exceeds 16", (unsigned long long)k, segs[k].bucket_len);
        next += segs[k].n_obj;
        sh.no_fill |= (segs[k].flags & OVAL_DESC_NO_FILL) != 0;
    }
    if (next != n_obj)
        return fail(OVAL_E_LAYOUT, "the segments cover %llu records, the table holds %llu", (unsigned long long)next,
                    (unsigned long long)n_obj);
    // units and key range, record range by record range on the helper threads (a table may be one segment)
    constexpr uint64_t kRecsPerTask = 128u << 10;
    const uint64_t n_tasks = (n_obj + kRecsPerTask - 1) / kRecsPerTask;
    std::vector<uint64_t> task_units(n_tasks, 0), task_bad(n_tasks, UINT64_MAX);
    TaskPool::get().run(n_tasks, [&](size_t t) {
        const uint64_t begin = t * kRecsPerTask, end = begin + kRecsPerTask < n_obj ? begin + kRecsPerTask : n_obj;
        uint64_t lo = 0, hi = n_segs - 1;                        // the segment that holds record `begin`
        while (lo < hi) {
            const uint64_t mid = (lo + hi + 1) / 2;
            if (segs[mid].first_obj <= begin) lo = mid; else hi = mid - 1;
        }
        uint64_t k = lo, units = 0;
        for (uint64_t i = begin; i < end; ++i) {
            while (i >= (uint64_t)segs[k].first_obj + segs[k].n_obj) ++k;
            units += recs[i].n_units;
            if (segs[k].key_id_base + recs[i].obj_id >= (1ull << 40) && task_bad[t] == UINT64_MAX) task_bad[t] = i;
        }
        task_units[t] = units;
    });
    for (uint64_t k = 0; k < n_segs; ++k)
        if (segs[k].key_id_base >= (1ull << 40))
            return fail(OVAL_E_LAYOUT, "segment %llu: key range base %llu needs more than 10 hex digits", (unsigned long long)k,
                        (unsigned long long)segs[k].key_id_base);
    for (uint64_t t = 0; t < n_tasks; ++t) {
        if (task_bad[t] != UINT64_MAX)
            return fail(OVAL_E_LAYOUT, "record %llu: its object ID needs more than 10 hex digits, the key would exceed 12 bytes",
                        (unsigned long long)task_bad[t]);
        sh.total_units += task_units[t];
    }
    if (sh.total_units >= 0xFFFFFF00ull)
        return fail(OVAL_E_LAYOUT, "batch spans %llu bytes; the limit is just under 1 TiB",
                    (unsigned long long)(sh.total_units * OVAL_UNIT_SIZE));
    sh.span_bytes = sh.total_units * OVAL_UNIT_SIZE;
    // staging and device scratch: records | segments (16-byte aligned) | block sums
    const uint64_t recs_bytes = (n_obj * sizeof(oval_obj_rec) + 15) & ~15ull;
    const uint64_t segs_bytes = n_segs * sizeof(oval_obj_segment);
    const uint64_t sums_bytes = (uint64_t)oval::compact_blocks((uint32_t)n_obj) * 8;
    sh.stage_bytes = recs_bytes + segs_bytes;
    sh.compact_bytes = recs_bytes + segs_bytes + sums_bytes;
    hipStream_t build_stream = nullptr;
    int rc = layout_prepare(ctx, l, sh, (flags & OVAL_RESET_AHEAD) != 0, &build_stream);
    if (rc) return rc;
    if (n_obj) {
        char* stage = static_cast<char*>(l->stage_host);
        oval_internal_parallel_copy(stage, recs, n_obj * sizeof(oval_obj_rec));
        std::memcpy(stage + recs_bytes, segs, segs_bytes);
        char* dev = static_cast<char*>(l->compact_dev);
        HIP_TRY(hipMemcpyAsync(dev, stage, recs_bytes + segs_bytes, hipMemcpyHostToDevice, build_stream));
        HIP_TRY(oval::launch_expand_descs(build_stream, dev, (uint32_t)n_obj, dev + recs_bytes, (uint32_t)n_segs,
                                          reinterpret_cast<unsigned long long*>(dev + recs_bytes + segs_bytes), l->descs_dev));
    }
    return layout_finish(l, sh, build_stream);
} catch (...) {
    return oval_internal_caught();
}

int oval_layout_read_descs(oval_ctx* ctx, const oval_layout* l, uint64_t first, uint64_t n, oval_obj_desc* out_host)
try {
    if (!ctx || !l || (!out_host && n)) return fail(OVAL_E_INVALID, "NULL argument");
    if (first > l->n_obj || n > l->n_obj - first)
        return fail(OVAL_E_RANGE, "descriptors [%llu, %llu) of a table of %llu", (unsigned long long)first,
                    (unsigned long long)(first + n), (unsigned long long)l->n_obj);
    if (n == 0) return OVAL_OK;
    if (l->built_ev) HIP_TRY(hipEventSynchronize(l->built_ev));
    HIP_TRY(hipMemcpy(out_host, static_cast<const oval_obj_desc*>(l->descs_dev) + first, n * sizeof(oval_obj_desc),
                      hipMemcpyDeviceToHost));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

// A pass on another stream than the one the tables were built on waits for them.
static int await_tables(oval_ctx* ctx, const oval_layout* l)
{
    if (l->built_ev && l->built_on != ctx->stream) HIP_TRY(hipStreamWaitEvent(ctx->stream, l->built_ev, 0));
    return OVAL_OK;
}

// After a pass over `l` has been issued on the context's stream.
static int note_use(oval_ctx* ctx, const oval_layout* l)
{
    if (!l->track_use || !l->used_ev) return OVAL_OK;
    HIP_TRY(hipEventRecord(l->used_ev, ctx->stream));
    l->used = true;
    return OVAL_OK;
}

int oval_layout_create(oval_ctx* ctx, const oval_obj_desc* descs, uint64_t n_obj, oval_layout** out)
try {
    if (!ctx || !out || (!descs && n_obj)) return fail(OVAL_E_INVALID, "NULL argument");
    *out = nullptr;
    oval_layout* l = new (std::nothrow) oval_layout;
    if (!l) return fail(OVAL_E_NOMEM, "out of host memory");
    l->device = ctx->device;
    int rc = oval_layout_reset(ctx, l, descs, n_obj);
    if (rc) {
        if (l->descs_dev) (void)hipFree(l->descs_dev);
        if (l->chunk_first_dev) (void)hipFree(l->chunk_first_dev);
        if (l->chunk_desc_dev) (void)hipFree(l->chunk_desc_dev);
        if (l->stage_host) (void)hipHostFree(l->stage_host);
        if (l->compact_dev) (void)hipFree(l->compact_dev);
        if (l->built_ev) (void)hipEventDestroy(l->built_ev);
        if (l->used_ev) (void)hipEventDestroy(l->used_ev);
        delete l;
        return rc;
    }
    *out = l;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_layout_destroy(oval_ctx* ctx, oval_layout* l)
try {
    if (!l) return OVAL_OK;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    if (l->built_ev) (void)hipEventSynchronize(l->built_ev);   // a build on the copy stream still writes the tables
    if (l->descs_dev) (void)hipFree(l->descs_dev);
    if (l->chunk_first_dev) (void)hipFree(l->chunk_first_dev);
    if (l->chunk_desc_dev) (void)hipFree(l->chunk_desc_dev);
    if (l->stage_host) (void)hipHostFree(l->stage_host);
    if (l->compact_dev) (void)hipFree(l->compact_dev);
    if (l->built_ev) (void)hipEventDestroy(l->built_ev);
    if (l->used_ev) (void)hipEventDestroy(l->used_ev);
    delete l;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_layout_launches(const oval_layout* l, uint32_t* fill_launches, uint32_t* validate_launches)
try {
    if (!l) return fail(OVAL_E_INVALID, "layout is NULL");
    const uint32_t n = l->n_tiles;
    if (fill_launches) {
        const uint32_t per = oval::fill_tiles_per_launch(view_of(l), n);
        *fill_launches = n && l->n_obj ? (n + per - 1) / per : 0;
    }
    if (validate_launches)
        *validate_launches = n && l->n_obj ? (n + oval::kMaxTilesPerLaunch - 1) / oval::kMaxTilesPerLaunch : 0;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_layout_info(const oval_layout* l, uint64_t* n_obj, uint64_t* total_units, uint64_t* span_bytes)
try {
    if (!l) return fail(OVAL_E_INVALID, "layout is NULL");
    if (n_obj) *n_obj = l->n_obj;
    if (total_units) *total_units = l->total_units;
    if (span_bytes) *span_bytes = l->span_bytes;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

static int check_body(const oval_ctx* ctx, const oval_layout* l, const void* body, uint64_t body_bytes)
{
    if (!ctx || !l) return fail(OVAL_E_INVALID, "NULL argument");
    if (l->device != ctx->device) return fail(OVAL_E_INVALID, "layout belongs to device %d, context to %d", l->device, ctx->device);
    if (l->span_bytes && !body) return fail(OVAL_E_INVALID, "body is NULL");
    if (((uintptr_t)body) % 256) return fail(OVAL_E_INVALID, "body pointer %p is not 256-byte aligned", body);
    if (body_bytes < l->span_bytes)
        return fail(OVAL_E_RANGE, "body holds %llu bytes but the layout spans %llu",
                    (unsigned long long)body_bytes, (unsigned long long)l->span_bytes);
    return OVAL_OK;
}

int oval_fill(oval_ctx* ctx, const oval_layout* l, void* body_dev, uint64_t body_bytes)
try {
    int rc = check_body(ctx, l, body_dev, body_bytes);
    if (rc) return rc;
    if (l->no_fill)
        return fail(OVAL_E_FIELD, "the layout holds an object whose names do not format to 28 bytes; "
                                  "the reference's Generate fails for it (pattern.go:87-89)");
    if ((rc = await_tables(ctx, l))) return rc;
    HIP_TRY(oval::launch_fill(ctx->stream, cfg_of(ctx), view_of(l), body_dev, 0, tiles_of(l)));
    return note_use(ctx, l);
} catch (...) {
    return oval_internal_caught();
}

int oval_validate(oval_ctx* ctx, const oval_layout* l, const void* body_dev, uint64_t body_bytes,
                  oval_obj_result* results_dev, uint64_t* counters_dev)
try {
    int rc = check_body(ctx, l, body_dev, body_bytes);
    if (rc) return rc;
    if (!results_dev && l->n_obj) return fail(OVAL_E_INVALID, "results_dev is NULL");
    if (((uintptr_t)results_dev) % 16) return fail(OVAL_E_INVALID, "results_dev is not 16-byte aligned");
    if (((uintptr_t)counters_dev) % 8) return fail(OVAL_E_INVALID, "counters_dev is not 8-byte aligned");
    const oval::LayoutView v = view_of(l);
    if ((rc = await_tables(ctx, l))) return rc;
    HIP_TRY(oval::launch_init_results(ctx->stream, v, results_dev, (unsigned long long*)counters_dev));
    HIP_TRY(oval::launch_validate(ctx->stream, cfg_of(ctx), v, body_dev, 0, tiles_of(l), results_dev,
                                  (unsigned long long*)counters_dev, ctx->tallies_dev));
    return note_use(ctx, l);
} catch (...) {
    return oval_internal_caught();
}

int oval_validate_noalloc(oval_ctx* ctx, const oval_obj_desc* descs_dev, uint64_t n_obj, uint64_t span_bytes,
                          const void* body_dev, uint64_t body_bytes, oval_obj_result* results_dev,
                          uint64_t* counters_dev)
try {
    if (!ctx) return fail(OVAL_E_INVALID, "ctx is NULL");
    if (n_obj > kMaxObjects) return fail(OVAL_E_LAYOUT, "too many objects (%llu): a batch holds at most 2^30", (unsigned long long)n_obj);
    if (span_bytes % OVAL_UNIT_SIZE) return fail(OVAL_E_INVALID, "span_bytes %llu is not a multiple of 256",
                                                 (unsigned long long)span_bytes);
    if (span_bytes > OVAL_NOALLOC_MAX_SPAN)
        return fail(OVAL_E_RANGE, "span of %llu bytes is beyond the %llu bytes the static chunk table covers; "
                                  "use oval_layout_create + oval_validate",
                    (unsigned long long)span_bytes, (unsigned long long)OVAL_NOALLOC_MAX_SPAN);
    if (n_obj && (!descs_dev || !results_dev)) return fail(OVAL_E_INVALID, "NULL argument");
    if (span_bytes && !body_dev) return fail(OVAL_E_INVALID, "body is NULL");
    if (((uintptr_t)descs_dev) % 64) return fail(OVAL_E_INVALID, "descs_dev is not 64-byte aligned");
    if (((uintptr_t)body_dev) % 256) return fail(OVAL_E_INVALID, "body pointer %p is not 256-byte aligned", body_dev);
    if (((uintptr_t)results_dev) % 16) return fail(OVAL_E_INVALID, "results_dev is not 16-byte aligned");
    if (((uintptr_t)counters_dev) % 8) return fail(OVAL_E_INVALID, "counters_dev is not 8-byte aligned");
    if (body_bytes < span_bytes)
        return fail(OVAL_E_RANGE, "body holds %llu bytes but the span is %llu", (unsigned long long)body_bytes,
                    (unsigned long long)span_bytes);
    HIP_TRY(hipSetDevice(ctx->device));                          // (the calling thread may last have used another GPU)
    HIP_TRY(oval::launch_validate_noalloc(ctx->stream, ctx->device, cfg_of(ctx), descs_dev, (uint32_t)n_obj,
                                          (uint32_t)(span_bytes / OVAL_UNIT_SIZE), body_dev, results_dev,
                                          (unsigned long long*)counters_dev));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_read_unit(oval_ctx* ctx, const void* body_dev, uint64_t byte_offset, uint8_t* out256)
try {
    if (!ctx || !body_dev || !out256) return fail(OVAL_E_INVALID, "NULL argument");
    if (byte_offset % OVAL_UNIT_SIZE) return fail(OVAL_E_INVALID, "byte_offset is not a multiple of 256");
    HIP_TRY(hipMemcpyAsync(out256, (const char*)body_dev + byte_offset, OVAL_UNIT_SIZE, hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

}  // extern "C"

// ---- bodies in host memory: stage through the device in slices -------------
// A slice is a whole number of tiles (64 KiB each) and lives in one slot of a
// small ring of device memory owned by the context; the layout, the device-side
// results and counters and (for pageable bodies) a pinned bounce ring are owned
// by the context too and only ever grow, so a call allocates nothing once the
// context has seen a batch of its size.  The kernel for slice k runs on the main
// stream while the copies of its neighbours run on the copy stream.  A kernel
// is handed `slot - slice_offset` as the body address: it addresses the body
// by absolute offset, and exactly the offsets of its slice land in the slot.
//
// A pageable body cannot be the end point of an asynchronous copy (the runtime
// stages it through a small buffer of its own, synchronously: 17-28 GB/s), so
// it moves between the caller's memory and a pinned bounce slot with a memcpy
// spread over a few host threads while the copy engine works on the neighbouring
// slot.
namespace {
constexpr uint64_t kTileBytes = (uint64_t)oval::kUnitsPerTile * OVAL_UNIT_SIZE;
constexpr uint32_t kSliceTiles = 512;   // 32 MiB
constexpr uint64_t kSliceBytes = kSliceTiles * kTileBytes;
constexpr size_t kMaxRunsPerSlice = 64;  // more separate bodies than this in a slice: the slice round-trips instead
constexpr size_t kCopySegBytes = 1u << 20;

struct CopySeg {
    void* dst;
    const void* src;
    size_t n;
};

// memcpy of a list of segments spread over a few threads; the caller takes part.
void copy_segments(const CopySeg* segs, size_t n)
{
    size_t bytes = 0;
    for (size_t i = 0; i < n; ++i) bytes += segs[i].n;
    if (n <= 1 || bytes < (4u << 20)) {
        for (size_t i = 0; i < n; ++i) std::memcpy(segs[i].dst, segs[i].src, segs[i].n);
        return;
    }
    TaskPool::get().run(n, [segs](size_t i) { std::memcpy(segs[i].dst, segs[i].src, segs[i].n); });
}

void add_segs(std::vector<CopySeg>& out, char* dst, const char* src, uint64_t n)
{
    for (uint64_t o = 0; o < n; o += kCopySegBytes)
        out.push_back(CopySeg{dst + o, src + o, (size_t)(n - o < kCopySegBytes ? n - o : kCopySegBytes)});
}

// true when `p` is host memory the copy engines can reach directly
bool is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof a);
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();         // plain malloc memory is reported as an invalid value
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// The byte ranges of a slice that belong to an object, neighbours merged; walks
// the (ascending) descriptor array once over all the slices of a call.
struct RunWalker {
    const oval_obj_desc* d;
    uint64_t n, i = 0;
    void runs(uint64_t b0, uint64_t b1, std::vector<std::pair<uint64_t, uint64_t>>& out)
    {
        out.clear();
        while (i < n && d[i].body_offset + (uint64_t)d[i].n_units * OVAL_UNIT_SIZE <= b0) ++i;
        for (uint64_t k = i; k < n && d[k].body_offset < b1; ++k) {
            const uint64_t s = d[k].body_offset > b0 ? d[k].body_offset : b0;
            uint64_t e = d[k].body_offset + (uint64_t)d[k].n_units * OVAL_UNIT_SIZE;
            if (e > b1) e = b1;
            if (e <= s) continue;
            if (!out.empty() && out.back().second == s) out.back().second = e;
            else out.emplace_back(s, e);
        }
    }
};

hipError_t ensure_ring(oval_ctx* c)
{
    if (c->ring_dev) return hipSuccess;
    hipError_t e = hipSuccess;
    for (int i = 0; i < oval_ctx::kRingSlots && e == hipSuccess; ++i) {
        if (!c->ring_ready[i]) e = hipEventCreateWithFlags(&c->ring_ready[i], hipEventDisableTiming);
        if (e == hipSuccess && !c->ring_free[i]) e = hipEventCreateWithFlags(&c->ring_free[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMalloc(&c->ring_dev, oval_ctx::kRingSlots * kSliceBytes);
    return e;
}

hipError_t ensure_bounce(oval_ctx* c)
{
    if (c->bounce_host) return hipSuccess;
    hipError_t e = hipSuccess;
    for (int i = 0; i < oval_ctx::kRingSlots && e == hipSuccess; ++i)
        if (!c->bounce_ev[i]) e = hipEventCreateWithFlags(&c->bounce_ev[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc(&c->bounce_host, oval_ctx::kRingSlots * kSliceBytes, hipHostMallocDefault);
    return e;
}

// the body address that makes absolute offset b0 fall on the start of `slot`
char* slot_base(const oval_ctx* c, uint32_t slot, uint64_t b0)
{
    return (char*)c->ring_dev + (uint64_t)slot * kSliceBytes - b0;
}

char* bounce_slot(const oval_ctx* c, uint32_t slot) { return (char*)c->bounce_host + (uint64_t)slot * kSliceBytes; }

// The context's own layout, reset to `descs`; checks the host body against it.
int host_layout_for(oval_ctx* ctx, const oval_obj_desc* descs, uint64_t n_obj, const void* body_host,
                    uint64_t body_bytes, oval_layout** out)
{
    int rc;
    if (!ctx->host_layout) {
        if ((rc = oval_layout_create(ctx, nullptr, 0, &ctx->host_layout))) return rc;
    }
    oval_layout* l = ctx->host_layout;
    if ((rc = oval_layout_reset(ctx, l, descs, n_obj))) return rc;
    if (body_bytes < l->span_bytes || (!body_host && l->span_bytes))
        return fail(OVAL_E_RANGE, "host body holds %llu bytes but the layout spans %llu",
                    (unsigned long long)body_bytes, (unsigned long long)l->span_bytes);
    *out = l;
    return OVAL_OK;
}

}  // namespace

int oval_internal_pinned_alloc_coherent(oval_ctx* ctx, size_t bytes, void** hptr_out)
{
    if (!ctx || !hptr_out) return fail(OVAL_E_INVALID, "NULL argument");
    *hptr_out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipHostMalloc(hptr_out, bytes ? bytes : 1, hipHostMallocCoherent | hipHostMallocMapped));
    return OVAL_OK;
}

void oval_internal_parallel_for(size_t n, const std::function<void(size_t)>& fn) { TaskPool::get().run(n, fn); }

void oval_internal_parallel_copy(void* dst, const void* src, size_t n)
{
    std::vector<CopySeg> segs;
    add_segs(segs, (char*)dst, (const char*)src, n);
    copy_segments(segs.data(), segs.size());
}

uint32_t oval_internal_host_slices(uint64_t span_bytes)
{
    return (uint32_t)((span_bytes + kSliceBytes - 1) / kSliceBytes);
}

uint64_t oval_internal_host_slice_bytes(void) { return kSliceBytes; }

// oval_fill_batch_host, optionally without the final wait: `landed[k]` (if not
// NULL; then body_host must be pinned) is recorded on the copy stream when slice
// k, bytes [k * 32 MiB, (k + 1) * 32 MiB) of the body, has arrived in host memory.
int oval_internal_fill_host(oval_ctx* ctx, const oval_obj_desc* descs, uint64_t n_obj, void* body_host,
                            uint64_t body_bytes, hipEvent_t* landed, uint32_t n_landed, bool wait)
{
    if (!ctx) return fail(OVAL_E_INVALID, "ctx is NULL");
    oval_layout* l = nullptr;
    int rc = host_layout_for(ctx, descs, n_obj, body_host, body_bytes, &l);
    if (rc) return rc;
    if (l->no_fill)
        return fail(OVAL_E_FIELD, "the batch holds an object whose names do not format to 28 bytes; "
                                  "the reference's Generate fails for it (pattern.go:87-89)");
    const uint32_t n_tiles = tiles_of(l);
    if (landed && n_landed < oval_internal_host_slices(l->span_bytes))
        return fail(OVAL_E_INVALID, "one event per 32 MiB slice is needed");
    const bool pinned = landed ? true : (l->span_bytes == 0 || is_pinned(body_host));
    hipError_t e = ensure_ring(ctx);
    if (e == hipSuccess && !pinned) e = ensure_bounce(ctx);
    // Only bytes that belong to an object may change in the caller's buffer:
    // the kernel writes nothing else into the slot, so gaps are not copied back.
    const bool contiguous = l->total_units * OVAL_UNIT_SIZE == l->span_bytes;
    const oval::LayoutView v = view_of(l);
    RunWalker walker{descs, n_obj};
    std::vector<std::pair<uint64_t, uint64_t>> runs;
    constexpr uint32_t kSlots = oval_ctx::kRingSlots, kLag = 2;
    struct Drain { uint64_t b0; std::vector<CopySeg> segs; };
    Drain drains[kSlots];
    auto drain = [&](uint32_t k) {                              // pageable: bounce slot -> caller's memory
        Drain& d = drains[k % kSlots];
        hipError_t de = oval_internal_wait_event(ctx->bounce_ev[k % kSlots], false);
        if (de == hipSuccess) copy_segments(d.segs.data(), d.segs.size());
        return de;
    };
    uint32_t k = 0;
    for (uint32_t t0 = 0; t0 < n_tiles && e == hipSuccess; t0 += kSliceTiles, ++k) {
        const uint32_t t1 = t0 + kSliceTiles < n_tiles ? t0 + kSliceTiles : n_tiles;
        const uint32_t slot = k % kSlots;
        const uint64_t b0 = (uint64_t)t0 * kTileBytes;
        uint64_t b1 = (uint64_t)t1 * kTileBytes;
        if (b1 > l->span_bytes) b1 = l->span_bytes;
        if (contiguous) { runs.clear(); if (b1 > b0) runs.emplace_back(b0, b1); }
        else walker.runs(b0, b1, runs);
        const bool round_trip = pinned && runs.size() > kMaxRunsPerSlice;
        char* const base = slot_base(ctx, slot, b0);
        const hipStream_t cs = (k & 1) ? ctx->copy_stream2 : ctx->copy_stream;
        // the slot's previous slice must have left for the host
        if (k >= kSlots) e = hipStreamWaitEvent(ctx->stream, ctx->ring_free[slot], 0);
        if (e == hipSuccess && round_trip)                      // gaps travel there and back unchanged
            e = hipMemcpyAsync(base + b0, (const char*)body_host + b0, b1 - b0, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = oval::launch_fill(ctx->stream, cfg_of(ctx), v, base, t0, t1);
        if (e == hipSuccess) e = hipEventRecord(ctx->ring_ready[slot], ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(cs, ctx->ring_ready[slot], 0);
        if (pinned) {
            if (round_trip) { runs.clear(); runs.emplace_back(b0, b1); }
            for (size_t r = 0; r < runs.size() && e == hipSuccess; ++r)
                e = hipMemcpyAsync((char*)body_host + runs[r].first, base + runs[r].first,
                                   runs[r].second - runs[r].first, hipMemcpyDeviceToHost, cs);
        } else {
            if (e == hipSuccess && b1 > b0)
                e = hipMemcpyAsync(bounce_slot(ctx, slot), base + b0, b1 - b0, hipMemcpyDeviceToHost, cs);
            if (e == hipSuccess) e = hipEventRecord(ctx->bounce_ev[slot], cs);
            Drain& d = drains[slot];
            d.b0 = b0;
            d.segs.clear();
            for (const auto& r : runs)
                add_segs(d.segs, (char*)body_host + r.first, bounce_slot(ctx, slot) + (r.first - b0), r.second - r.first);
        }
        if (e == hipSuccess) e = hipEventRecord(ctx->ring_free[slot], cs);
        if (e == hipSuccess && landed) e = hipEventRecord(landed[k], cs);
        if (e == hipSuccess && !pinned && k >= kLag) e = drain(k - kLag);
    }
    if (!pinned)
        for (uint32_t q = k > kLag ? k - kLag : 0; q < k && e == hipSuccess; ++q) e = drain(q);
    if (wait || e != hipSuccess) {
        const hipError_t e1 = oval_internal_wait_stream(ctx->stream);
        const hipError_t e2 = oval_internal_wait_stream(ctx->copy_stream);
        const hipError_t e3 = oval_internal_wait_stream(ctx->copy_stream2);
        if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3;
    }
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? OVAL_E_NOMEM : OVAL_E_HIP, "fill_batch_host: %s", oval_internal_wait_error(e));
    return OVAL_OK;
}

extern "C" {

int oval_fill_batch_host(oval_ctx* ctx, const oval_obj_desc* descs, uint64_t n_obj, void* body_host,
                         uint64_t body_bytes)
try {
    return oval_internal_fill_host(ctx, descs, n_obj, body_host, body_bytes, nullptr, 0, true);
} catch (...) {
    return oval_internal_caught();
}

int oval_validate_batch_host(oval_ctx* ctx, const oval_obj_desc* descs, uint64_t n_obj,
                             const void* body_host, uint64_t body_bytes, oval_obj_result* results_host,
                             uint64_t* counters_host)
try {
    if (!ctx || (!results_host && n_obj) || (!descs && n_obj)) return fail(OVAL_E_INVALID, "NULL argument");
    // Where the bodies end follows from the last descriptor (the table ascends;
    // the layout build below checks that it does).  The first slices start their
    // way to the device before the descriptor table is even looked at: scanning
    // and staging a million descriptors takes the host several milliseconds.
    const uint64_t span = n_obj ? descs[n_obj - 1].body_offset + (uint64_t)descs[n_obj - 1].n_units * OVAL_UNIT_SIZE : 0;
    if (body_bytes < span || (!body_host && span))
        return fail(OVAL_E_RANGE, "host body holds %llu bytes but the layout spans %llu",
                    (unsigned long long)body_bytes, (unsigned long long)span);
    if (span / OVAL_UNIT_SIZE >= 0xFFFFFF00ull)
        return fail(OVAL_E_LAYOUT, "batch spans %llu bytes; the limit is just under 1 TiB", (unsigned long long)span);
    const bool pinned = span == 0 || is_pinned(body_host);
    hipError_t e = ensure_ring(ctx);
    if (e == hipSuccess && !pinned) e = ensure_bounce(ctx);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? OVAL_E_NOMEM : OVAL_E_HIP, "validate_batch_host: %s", hipGetErrorString(e));
    constexpr uint32_t kSlots = oval_ctx::kRingSlots;
    const uint32_t n_slices = oval_internal_host_slices(span);
    std::vector<CopySeg> segs;
    // slice k on its way into ring slot k % kSlots
    auto upload = [&](uint32_t k) {
        const uint32_t slot = k % kSlots;
        const uint64_t b0 = (uint64_t)k * kSliceBytes;
        const uint64_t b1 = b0 + kSliceBytes < span ? b0 + kSliceBytes : span;
        const char* src = (const char*)body_host + b0;
        const hipStream_t cs = (k & 1) ? ctx->copy_stream2 : ctx->copy_stream;
        hipError_t ue = hipSuccess;
        if (!pinned) {
            // the bounce slot's previous slice must have left for the device
            if (k >= kSlots) ue = oval_internal_wait_event(ctx->bounce_ev[slot], false);
            if (ue != hipSuccess) return ue;
            segs.clear();
            add_segs(segs, bounce_slot(ctx, slot), src, b1 - b0);
            copy_segments(segs.data(), segs.size());
            src = bounce_slot(ctx, slot);
        }
        // the slot's previous slice must have been validated
        if (k >= kSlots) ue = hipStreamWaitEvent(cs, ctx->ring_free[slot], 0);
        if (ue == hipSuccess)
            ue = hipMemcpyAsync(slot_base(ctx, slot, b0) + b0, src, b1 - b0, hipMemcpyHostToDevice, cs);
        if (ue == hipSuccess && !pinned) ue = hipEventRecord(ctx->bounce_ev[slot], cs);
        if (ue == hipSuccess) ue = hipEventRecord(ctx->ring_ready[slot], cs);
        return ue;
    };
    uint32_t uploaded = 0;
    for (; uploaded < n_slices && uploaded < kSlots && e == hipSuccess; ++uploaded) e = upload(uploaded);
    oval_layout* l = nullptr;
    int rc = e == hipSuccess ? host_layout_for(ctx, descs, n_obj, body_host, body_bytes, &l) : OVAL_OK;
    if (rc || (l && l->span_bytes != span)) {
        (void)oval_internal_wait_stream(ctx->copy_stream);
        (void)oval_internal_wait_stream(ctx->copy_stream2);
        return rc ? rc : fail(OVAL_E_LAYOUT, "descriptor table does not end where its last entry does");
    }
    if (e == hipSuccess && (n_obj > ctx->host_res_cap || !ctx->host_res_dev)) {
        if (ctx->host_res_dev) {
            e = oval_internal_wait_stream(ctx->stream);          // a pass may still write the old array
            if (e == hipSuccess) (void)hipFree(ctx->host_res_dev);   // (one that never ends keeps it: leaked, not freed under it)
        }
        ctx->host_res_dev = nullptr; ctx->host_res_cap = 0;
        const uint64_t cap = n_obj < 1024 ? 1024 : n_obj + n_obj / 8;
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->host_res_dev, cap * sizeof(oval_obj_result));
        if (e == hipSuccess) ctx->host_res_cap = cap;
    }
    if (e == hipSuccess && !ctx->host_cnt_dev) e = hipMalloc((void**)&ctx->host_cnt_dev, OVAL_NUM_COUNTERS * 8);
    oval_obj_result* const res_dev = ctx->host_res_dev;
    unsigned long long* const cnt_dev = ctx->host_cnt_dev;
    if (e == hipSuccess) e = hipMemsetAsync(cnt_dev, 0, OVAL_NUM_COUNTERS * 8, ctx->stream);
    if (e == hipSuccess) {
        const oval::LayoutView v = view_of(l);
        e = oval::launch_init_results(ctx->stream, v, res_dev, cnt_dev);
        for (uint32_t k = 0; k < n_slices && e == hipSuccess; ++k) {
            if (k >= uploaded) { e = upload(k); ++uploaded; }
            const uint32_t slot = k % kSlots;
            const uint32_t t0 = k * kSliceTiles;
            const uint32_t t1 = t0 + kSliceTiles < l->n_tiles ? t0 + kSliceTiles : l->n_tiles;
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->ring_ready[slot], 0);
            if (e == hipSuccess)
                e = oval::launch_validate(ctx->stream, cfg_of(ctx), v, slot_base(ctx, slot, (uint64_t)k * kSliceBytes), t0, t1,
                                          res_dev, cnt_dev, ctx->tallies_dev);
            if (e == hipSuccess) e = hipEventRecord(ctx->ring_free[slot], ctx->stream);
        }
    }
    if (e == hipSuccess && n_obj)
        e = hipMemcpyAsync(results_host, res_dev, n_obj * sizeof(oval_obj_result), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && counters_host)
        e = hipMemcpyAsync(counters_host, cnt_dev, OVAL_NUM_COUNTERS * 8, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e1 = oval_internal_wait_stream(ctx->stream);
    const hipError_t e2 = oval_internal_wait_stream(ctx->copy_stream);
    const hipError_t e3 = oval_internal_wait_stream(ctx->copy_stream2);
    if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? OVAL_E_NOMEM : OVAL_E_HIP, "validate_batch_host: %s", oval_internal_wait_error(e));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

}  // extern "C"

extern "C" {

// ---- timing -----------------------------------------------------------------
int oval_timer_create(oval_ctx* ctx, oval_timer** out)
try {
    if (!ctx || !out) return fail(OVAL_E_INVALID, "NULL argument");
    *out = nullptr;
    oval_timer* t = new (std::nothrow) oval_timer;
    if (!t) return fail(OVAL_E_NOMEM, "out of host memory");
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = hipEventCreate(&t->start);
    if (e == hipSuccess) e = hipEventCreate(&t->stop);
    if (e != hipSuccess) { delete t; return fail(OVAL_E_HIP, "hipEventCreate: %s", hipGetErrorString(e)); }
    *out = t;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_timer_destroy(oval_timer* t)
try {
    if (!t) return OVAL_OK;
    if (t->start) (void)hipEventDestroy(t->start);
    if (t->stop) (void)hipEventDestroy(t->stop);
    delete t;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_timer_start(oval_ctx* ctx, oval_timer* t)
try {
    if (!ctx || !t) return fail(OVAL_E_INVALID, "NULL argument");
    HIP_TRY(hipEventRecord(t->start, ctx->stream));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_timer_stop(oval_ctx* ctx, oval_timer* t)
try {
    if (!ctx || !t) return fail(OVAL_E_INVALID, "NULL argument");
    HIP_TRY(hipEventRecord(t->stop, ctx->stream));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_timer_elapsed_ms(oval_timer* t, float* ms_out)
try {
    if (!t || !ms_out) return fail(OVAL_E_INVALID, "NULL argument");
    HIP_TRY(hipEventSynchronize(t->stop));
    HIP_TRY(hipEventElapsedTime(ms_out, t->start, t->stop));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

}  // extern "C"
