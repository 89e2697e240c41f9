// oval_pattern.cpp — host-side mirror of internal/pattern and internal/object
// (see oval_pattern.hpp).  Every byte Generate returns is written by the fill
// kernel and every verdict Valid returns is taken by the validate kernel; the
// host only formats names, moves bytes and words the reference's error text.
#include "oval_pattern.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <random>
#include <thread>

#include "../../include/oval_hip.h"
#include "../../include/oval_pattern.h"
#include "oval_gostr.h"
#include "oval_trimspace.h"

namespace oval {

// ---------------------------------------------------------------------------
// internal/object
// ---------------------------------------------------------------------------
namespace object {

std::string generateKey(int64_t objID)
{
    // fmt.Sprintf("%s%010x", KeyShortPrefix, objID): the sign, if any, is part
    // of the ten columns
    char buf[40];
    if (objID < 0) {
        const unsigned long long mag = (unsigned long long)(-(objID + 1)) + 1ull;
        std::snprintf(buf, sizeof buf, "%s-%09llx", KeyShortPrefix, mag);
    } else {
        std::snprintf(buf, sizeof buf, "%s%010llx", KeyShortPrefix, (unsigned long long)objID);
    }
    return buf;
}

Object NewObject(int64_t objID)
{
    Object o;
    o.Key = generateKey(objID);
    return o;
}

}  // namespace object

namespace pattern {

// ---------------------------------------------------------------------------
// engine: one context, one reusable layout, growable staging buffers.  All
// device work goes through run_jobs_locked(), which takes any number of
// objects at once: the scalar API hands it one job, GenerateBatch/ValidBatch
// and the micro-batcher hand it many.
// ---------------------------------------------------------------------------
namespace {

struct Job {
    bool is_fill = true;
    oval_obj_desc desc{};          // body_offset is assigned by the runner
    uint8_t* out = nullptr;        // fill: n_units * 256 bytes of host memory to receive the body
    const uint8_t* in = nullptr;   // validate: n_units * 256 bytes of host memory holding the body
    oval_obj_result res{};         // validate: the device's verdict
    bool failed = false;           // the engine failed for the batch this job was in
    bool done = false;             // micro-batcher: completion flag ...
    std::condition_variable* waiter = nullptr;   // ... and who sleeps on it (only that caller is woken)
};

struct Engine {
    std::mutex mu;                 // serialises use of ctx/layout/staging
    oval_ctx* ctx = nullptr;
    oval_layout* layout = nullptr;
    void* dev_body = nullptr;
    size_t dev_cap = 0;
    uint8_t* pinned = nullptr;
    size_t pinned_cap = 0;
    oval_obj_result* res_dev = nullptr;
    oval_obj_result* res_host = nullptr;   // pinned
    size_t res_cap = 0;
    std::vector<oval_obj_desc> descs;
    std::string err;
    int device = 0;

    std::mutex state_mu;           // clock and random source
    int64_t clock = -1;
    uint64_t rng[4] = {0, 0, 0, 0};
    bool seeded = false;
};

Engine g_engine;

// The micro-batcher runs kBatchLanes launches at a time, each lane a thread
// with a context (stream, staging) of its own: lane 0 shares g_engine with the
// direct path, the others own one of these.
constexpr int kBatchLanes = 2;
Engine g_lane_engine[kBatchLanes - 1];

uint64_t splitmix64(uint64_t& x)
{
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void seed_locked(Engine& e, uint64_t seed)
{
    for (auto& w : e.rng) w = splitmix64(seed);
    e.seeded = true;
}

// xoshiro256** -> uniform double in [0,1) with 53 random bits, like rand.Float64()
double float64_locked(Engine& e)
{
    if (!e.seeded) {
        std::random_device rd;
        seed_locked(e, ((uint64_t)rd() << 32) ^ rd());
    }
    auto rotl = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
    uint64_t* s = e.rng;
    const uint64_t result = rotl(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return (double)(result >> 11) * 0x1.0p-53;
}

int engine_fail(Engine& e, int rc)
{
    e.err = oval_last_error();
    return rc;
}

int ensure_engine_locked(Engine& e, int device)
{
    if (e.ctx) return OVAL_OK;
    int rc = oval_ctx_create(device, nullptr, &e.ctx);
    if (rc) return engine_fail(e, rc);
    e.device = device;
    rc = oval_layout_create(e.ctx, nullptr, 0, &e.layout);
    if (rc) {
        engine_fail(e, rc);
        oval_ctx_destroy(e.ctx);
        e.ctx = nullptr;
        return rc;
    }
    return OVAL_OK;
}

int ensure_buffers_locked(Engine& e, size_t bytes, size_t n_results)
{
    if (bytes > e.dev_cap) {
        if (e.dev_body) { oval_ctx_sync(e.ctx); oval_dev_free(e.ctx, e.dev_body); e.dev_body = nullptr; e.dev_cap = 0; }
        size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
        cap = (cap + 65535) & ~(size_t)65535;
        int rc = oval_dev_alloc(e.ctx, cap, &e.dev_body);
        if (rc) return engine_fail(e, rc);
        e.dev_cap = cap;
    }
    if (bytes > e.pinned_cap) {
        if (e.pinned) { oval_ctx_sync(e.ctx); oval_pinned_free(e.ctx, e.pinned); e.pinned = nullptr; e.pinned_cap = 0; }
        size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
        int rc = oval_pinned_alloc(e.ctx, cap, (void**)&e.pinned);
        if (rc) return engine_fail(e, rc);
        e.pinned_cap = cap;
    }
    if (n_results > e.res_cap) {
        oval_ctx_sync(e.ctx);
        if (e.res_dev) oval_dev_free(e.ctx, e.res_dev);
        if (e.res_host) oval_pinned_free(e.ctx, e.res_host);
        e.res_dev = nullptr; e.res_host = nullptr; e.res_cap = 0;
        const size_t cap = n_results < 64 ? 64 : n_results * 2;
        int rc = oval_dev_alloc(e.ctx, cap * sizeof(oval_obj_result), (void**)&e.res_dev);
        if (!rc) rc = oval_pinned_alloc(e.ctx, cap * sizeof(oval_obj_result), (void**)&e.res_host);
        if (rc) return engine_fail(e, rc);
        e.res_cap = cap;
    }
    return OVAL_OK;
}

Error engine_error(const Engine& e)
{
    return Error::New("oval engine failure: " + e.err);
}

// One pass (fill or validate) over the jobs of one kind.
int run_pass_locked(Engine& e, Job* const* jobs, size_t n, bool fill)
{
    e.descs.clear();
    uint64_t off = 0;
    for (size_t i = 0; i < n; ++i) {
        Job* j = jobs[i];
        j->desc.body_offset = off;
        off += (uint64_t)j->desc.n_units * dataUnitSize;
        e.descs.push_back(j->desc);
    }
    if (off == 0) return OVAL_OK;
    int rc = ensure_buffers_locked(e, off, fill ? 0 : n);
    if (rc) return rc;
    if ((rc = oval_layout_reset(e.ctx, e.layout, e.descs.data(), n))) return engine_fail(e, rc);
    if (fill) {
        if ((rc = oval_fill(e.ctx, e.layout, e.dev_body, e.dev_cap))) return engine_fail(e, rc);
        if ((rc = oval_copy_d2h(e.ctx, e.pinned, e.dev_body, off))) return engine_fail(e, rc);
        if ((rc = oval_ctx_sync(e.ctx))) return engine_fail(e, rc);
        for (size_t i = 0; i < n; ++i)
            std::memcpy(jobs[i]->out, e.pinned + jobs[i]->desc.body_offset,
                        (size_t)jobs[i]->desc.n_units * dataUnitSize);
    } else {
        for (size_t i = 0; i < n; ++i)
            std::memcpy(e.pinned + jobs[i]->desc.body_offset, jobs[i]->in,
                        (size_t)jobs[i]->desc.n_units * dataUnitSize);
        if ((rc = oval_copy_h2d(e.ctx, e.dev_body, e.pinned, off))) return engine_fail(e, rc);
        if ((rc = oval_validate(e.ctx, e.layout, e.dev_body, e.dev_cap, e.res_dev, nullptr))) return engine_fail(e, rc);
        if ((rc = oval_copy_d2h(e.ctx, e.res_host, e.res_dev, n * sizeof(oval_obj_result)))) return engine_fail(e, rc);
        if ((rc = oval_ctx_sync(e.ctx))) return engine_fail(e, rc);
        for (size_t i = 0; i < n; ++i) jobs[i]->res = e.res_host[i];
    }
    return OVAL_OK;
}

// Runs a mixed set of jobs: all fills in one launch, all validates in another.
int run_jobs_locked(Engine& e, Job* const* jobs, size_t n)
{
    int rc = ensure_engine_locked(e, 0);
    std::vector<Job*> fills, vals;
    for (size_t i = 0; i < n; ++i) (jobs[i]->is_fill ? fills : vals).push_back(jobs[i]);
    if (!rc && !fills.empty()) rc = run_pass_locked(e, fills.data(), fills.size(), true);
    if (!rc && !vals.empty()) rc = run_pass_locked(e, vals.data(), vals.size(), false);
    if (rc)
        for (size_t i = 0; i < n; ++i) jobs[i]->failed = true;
    return rc;
}

// ---------------------------------------------------------------------------
// micro-batcher: coalesces concurrent single-object calls (one per worker
// thread, internal/runner/runner.go:186-214) into device batches.
// ---------------------------------------------------------------------------
struct Batcher {
    std::mutex mu;
    std::condition_variable work_cv;
    std::vector<Job*> queue;
    size_t queued_bytes = 0;
    bool enabled = false, stop = false;
    int window_us = 0;
    size_t max_bytes = 64u << 20;
    std::thread threads[kBatchLanes];
    uint64_t batches = 0, jobs = 0;   // statistics
};

Batcher g_batcher;

void batcher_loop(int lane)
{
    Batcher& b = g_batcher;
    Engine& engine = lane == 0 ? g_engine : g_lane_engine[lane - 1];
    std::unique_lock<std::mutex> lk(b.mu);
    for (;;) {
        b.work_cv.wait(lk, [&] { return b.stop || !b.queue.empty(); });
        if (b.stop && b.queue.empty()) return;
        // With window 0 a batch is whatever queued up while the previous one was
        // on the device (no added latency); a window > 0 also holds the door
        // open that long for callers arriving together.
        if (b.window_us > 0) {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(b.window_us);
            while (!b.stop && b.queued_bytes < b.max_bytes &&
                   b.work_cv.wait_until(lk, deadline) != std::cv_status::timeout) {
            }
        }
        std::vector<Job*> batch;
        batch.swap(b.queue);
        b.queued_bytes = 0;
        lk.unlock();
        {
            std::lock_guard<std::mutex> ek(engine.mu);
            int rc = ensure_engine_locked(engine, g_engine.device);
            if (!rc) rc = run_jobs_locked(engine, batch.data(), batch.size());
            else for (Job* j : batch) j->failed = true;
            if (rc && &engine != &g_engine) {
                std::lock_guard<std::mutex> sk(g_engine.state_mu);
                g_engine.err = engine.err;                      // where callers look for the text
            }
        }
        lk.lock();
        b.batches++;
        b.jobs += batch.size();
        for (Job* j : batch) {
            j->done = true;
            j->waiter->notify_one();
        }
    }
}

// Executes one job, through the batcher when it is enabled.
void execute(Job& job)
{
    Batcher& b = g_batcher;
    {
        std::unique_lock<std::mutex> lk(b.mu);
        if (b.enabled) {
            std::condition_variable mine;
            job.waiter = &mine;
            b.queue.push_back(&job);
            b.queued_bytes += (size_t)job.desc.n_units * dataUnitSize;
            b.work_cv.notify_one();
            mine.wait(lk, [&] { return job.done; });
            return;
        }
    }
    Job* p = &job;
    std::lock_guard<std::mutex> ek(g_engine.mu);
    run_jobs_locked(g_engine, &p, 1);
}

int64_t now_unix_micro()
{
    {
        std::lock_guard<std::mutex> lk(g_engine.state_mu);
        if (g_engine.clock >= 0) return g_engine.clock;
    }
    using namespace std::chrono;
    return duration_cast<microseconds>(system_clock::now().time_since_epoch()).count();
}

std::string truncate_bucket(const std::string& b)
{
    // pattern.go:74-76 and :119-121
    return b.size() > (size_t)object::MaxBucketNameLength ? b.substr(0, object::MaxBucketNameLength) : b;
}

// Builds the descriptor of a single-object call and reports the length the
// reference's formatting would give the unit (256 unless the names overflow).
int make_desc(const std::string& bucket16, const object::Object* obj, int64_t workerID, int64_t unixMicro,
              uint32_t firstUnit, uint32_t nUnits, oval_obj_desc* d, size_t* unit_len)
{
    const size_t bf = go_padded_len((const unsigned char*)bucket16.data(), bucket16.size(), object::MaxBucketNameLength);
    const size_t kf = go_padded_len((const unsigned char*)obj->Key.data(), obj->Key.size(), object::MaxKeyLength);
    *unit_len = bf + kf + dataUnitHeaderSizeWithoutBucketAndKey +
                (dataUnitSize - (object::MaxBucketNameLength + object::MaxKeyLength + dataUnitHeaderSizeWithoutBucketAndKey));
    return oval_desc_init(d, bucket16.data(), bucket16.size(), obj->Key.data(), obj->Key.size(), obj->WriteCount,
                          workerID, (uint64_t)unixMicro, 0, firstUnit, nUnits);
}

std::string trimmed(const uint8_t* p, int n)
{
    int a, b;
    trim_space([&](int i) { return (uint32_t)p[i]; }, n, a, b);
    return std::string((const char*)p + a, (size_t)(b - a));
}

// The text validDataUnit returns for a unit whose wrong fields are `mask`
// (pattern.go:139-177).
std::string unit_error_text(uint32_t mask, int64_t unitCount, int64_t workerID, const std::string& bucket16,
                            const object::Object* obj, const uint8_t* data)
{
    std::string msg;
    char num[128];
    auto u32at = [&](int off) {
        return (uint32_t)data[off] | ((uint32_t)data[off + 1] << 8) | ((uint32_t)data[off + 2] << 16) |
               ((uint32_t)data[off + 3] << 24);
    };
    if (mask & OVAL_BAD_BUCKET)
        msg += "- Bucket name is wrong. (expected = \"" + bucket16 + "\", actual = \"" +
               trimmed(data, object::MaxBucketNameLength) + "\")\n";
    if (mask & OVAL_BAD_KEY)
        msg += "- Key name is wrong. (expected = \"" + obj->Key + "\", actual = \"" +
               trimmed(data + object::MaxBucketNameLength, object::MaxKeyLength) + "\")\n";
    if (mask & OVAL_BAD_WRITECNT) {
        std::snprintf(num, sizeof num, "- WriteCount is wrong. (expected = \"%lld\", actual = \"%u\")\n",
                      (long long)obj->WriteCount, u32at(28));
        msg += num;
    }
    if (mask & OVAL_BAD_OFFSET) {
        std::snprintf(num, sizeof num, "- OffsetInObject is wrong. (expected = \"%lld\", actual = \"%u\")\n",
                      (long long)(unitCount * dataUnitSize), u32at(32));
        msg += num;
    }
    if (mask & OVAL_BAD_WORKER) {
        std::snprintf(num, sizeof num, "- WorkerID is wrong. (expected = \"%lld\", actual = \"%lld\")\n",
                      (long long)workerID, (long long)u32at(36));
        msg += num;
    }
    msg += dump(hexDump(data, dataUnitSize));
    return msg;
}

// math.Log / math.Log2 as the Go standard library computes them (its portable
// implementation, after FreeBSD's e_log.c), so that DecideSize truncates the
// same products.  Kept free of fused multiply-adds.
#pragma clang fp contract(off)
double go_log(double x)
{
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    int ki;
    double f1 = std::frexp(x, &ki);
    if (f1 < 1.41421356237309504880168872420969808 / 2) { f1 *= 2; ki--; }
    const double f = f1 - 1, k = (double)ki;
    const double s = f / (2 + f), s2 = s * s, s4 = s2 * s2;
    const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    const double R = t1 + t2, hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

double go_log2(double x)
{
    int e;
    const double frac = std::frexp(x, &e);
    if (frac == 0.5) return (double)(e - 1);
    return go_log(frac) * (1 / 0.693147180559945309417232121458176568) + (double)e;
}

}  // namespace

// ---------------------------------------------------------------------------
// engine control
// ---------------------------------------------------------------------------
int Init(int deviceID)
{
    std::lock_guard<std::mutex> lk(g_engine.mu);
    return ensure_engine_locked(g_engine, deviceID);
}

void SetMicroBatching(bool enable, int windowMicros, size_t maxBytes)
{
    Batcher& b = g_batcher;
    std::unique_lock<std::mutex> lk(b.mu);
    if (windowMicros >= 0) b.window_us = windowMicros;
    if (maxBytes > 0) b.max_bytes = maxBytes;
    if (enable && !b.enabled) {
        b.stop = false;
        b.enabled = true;
        for (int i = 0; i < kBatchLanes; ++i) b.threads[i] = std::thread(batcher_loop, i);
    } else if (!enable && b.enabled) {
        b.enabled = false;       // new callers take the direct path; queued ones are still served
        b.stop = true;
        b.work_cv.notify_all();
        std::thread ts[kBatchLanes];
        for (int i = 0; i < kBatchLanes; ++i) ts[i] = std::move(b.threads[i]);
        lk.unlock();
        for (std::thread& t : ts)
            if (t.joinable()) t.join();
    }
}

void MicroBatchingStats(uint64_t* batches, uint64_t* jobs)
{
    std::lock_guard<std::mutex> lk(g_batcher.mu);
    if (batches) *batches = g_batcher.batches;
    if (jobs) *jobs = g_batcher.jobs;
}

namespace {
void destroy_engine(Engine& e)
{
    std::lock_guard<std::mutex> lk(e.mu);
    if (!e.ctx) return;
    oval_ctx_sync(e.ctx);
    if (e.dev_body) oval_dev_free(e.ctx, e.dev_body);
    if (e.pinned) oval_pinned_free(e.ctx, e.pinned);
    if (e.res_dev) oval_dev_free(e.ctx, e.res_dev);
    if (e.res_host) oval_pinned_free(e.ctx, e.res_host);
    if (e.layout) oval_layout_destroy(e.ctx, e.layout);
    oval_ctx_destroy(e.ctx);
    e.ctx = nullptr; e.layout = nullptr; e.dev_body = nullptr; e.pinned = nullptr;
    e.res_dev = nullptr; e.res_host = nullptr; e.dev_cap = 0; e.pinned_cap = 0; e.res_cap = 0;
}
}  // namespace

void Shutdown()
{
    SetMicroBatching(false, -1, 0);
    for (Engine& e : g_lane_engine) destroy_engine(e);
    destroy_engine(g_engine);
}

void SetClock(int64_t unixMicro)
{
    std::lock_guard<std::mutex> lk(g_engine.state_mu);
    g_engine.clock = unixMicro;
}

void Seed(uint64_t seed)
{
    std::lock_guard<std::mutex> lk(g_engine.state_mu);
    seed_locked(g_engine, seed);
}

const std::string& LastEngineError() { return g_engine.err; }

// ---------------------------------------------------------------------------
// io
// ---------------------------------------------------------------------------
size_t BytesReader::Read(uint8_t* p, size_t n)
{
    const size_t left = n_ - pos_;
    if (n > left) n = left;
    if (n) std::memcpy(p, p_ + pos_, n);
    pos_ += n;
    return n;
}

// ---------------------------------------------------------------------------
// DecideSize — pattern.go:22-71
// ---------------------------------------------------------------------------
std::pair<int64_t, Error> DecideSizeFrom(int64_t minSize, int64_t maxSize, double y)
{
    if (minSize < dataUnitSize)
        return {0, Error::New("minSize should be larger than or equal to " + std::to_string(dataUnitSize))};
    if (minSize % dataUnitSize != 0)
        return {0, Error::New("minSize should be a multiple of " + std::to_string(dataUnitSize))};
    if (maxSize % dataUnitSize != 0)
        return {0, Error::New("maxSize should be a multiple of " + std::to_string(dataUnitSize))};
    if (maxSize < minSize)
        return {0, Error::New("maxSize should be larger than or equal to minSize")};
    const double x = -go_log2(1 - 0.5 * y);
    const double scaled = (double)((maxSize - minSize) / dataUnitSize + 1) * x;
    return {minSize + dataUnitSize * (int64_t)scaled, Error::Nil()};
}

std::pair<int64_t, Error> DecideSize(int64_t minSize, int64_t maxSize)
{
    double y;
    {
        std::lock_guard<std::mutex> lk(g_engine.state_mu);
        y = float64_locked(g_engine);
    }
    return DecideSizeFrom(minSize, maxSize, y);
}

// ---------------------------------------------------------------------------
// Generate — pattern.go:73-92, in three steps so that single calls and batches
// share them: prepare (host checks, descriptor), device pass, nothing to finish.
// ---------------------------------------------------------------------------
namespace {

// Returns nil and a job with n_units > 0 when there is device work to do; nil
// and n_units == 0 for an empty body; an error otherwise.
Error prepare_generate(int64_t dataSize, int64_t workerID, const std::string& bucketName,
                       const object::Object* obj, uint8_t* out, Job& job)
{
    const std::string bucket16 = truncate_bucket(bucketName);
    job.is_fill = true;
    job.out = out;
    job.desc.n_units = 0;
    if (dataSize < 0) return Error::New("negative data size");   // make([]byte, 0, dataSize) panics in Go
    const int64_t nUnits = dataSize / dataUnitSize;
    if (nUnits > 0xFFFFFFF0ll) return Error::New("data size beyond the engine's 1 TiB limit");
    size_t unit_len = 0;
    const int rc_desc = make_desc(bucket16, obj, workerID, now_unix_micro(), 0, (uint32_t)nUnits, &job.desc, &unit_len);
    // pattern.go:87-89: the length check that catches over-long fields and
    // sizes that are not a multiple of 256
    const int64_t produced = nUnits * (int64_t)unit_len;
    if (produced != dataSize) {
        char buf[128];
        std::snprintf(buf, sizeof buf, "generated data size is wrong. (expected: %lld, actual: %lld)",
                      (long long)dataSize, (long long)produced);
        job.desc.n_units = 0;
        return Error::New(buf);
    }
    if (nUnits > 0 && rc_desc) {
        job.desc.n_units = 0;
        return Error::New(std::string("oval engine failure: ") + oval_last_error());
    }
    return Error::Nil();
}

}  // namespace

Error GenerateInto(int64_t dataSize, int64_t workerID, const std::string& bucketName,
                   const object::Object* obj, uint8_t* out)
{
    Job job;
    Error err = prepare_generate(dataSize, workerID, bucketName, obj, out, job);
    if (!err.nil() || job.desc.n_units == 0) return err;
    execute(job);
    return job.failed ? engine_error(g_engine) : Error::Nil();
}

std::pair<std::vector<uint8_t>, Error> Generate(int64_t dataSize, int64_t workerID,
                                                const std::string& bucketName, const object::Object* obj)
{
    std::vector<uint8_t> data;
    if (dataSize > 0) data.resize((size_t)dataSize);
    Error err = GenerateInto(dataSize, workerID, bucketName, obj, data.data());
    if (!err.nil()) data.clear();
    return {std::move(data), err};
}

std::vector<Error> GenerateBatch(const std::vector<GenerateItem>& items)
{
    std::vector<Error> errs(items.size());
    std::vector<Job> jobs(items.size());
    std::vector<Job*> run;
    for (size_t i = 0; i < items.size(); ++i) {
        const GenerateItem& it = items[i];
        errs[i] = prepare_generate(it.dataSize, it.workerID, it.bucketName, it.obj, it.out, jobs[i]);
        if (errs[i].nil() && jobs[i].desc.n_units > 0) run.push_back(&jobs[i]);
    }
    if (!run.empty()) {
        std::lock_guard<std::mutex> ek(g_engine.mu);
        if (run_jobs_locked(g_engine, run.data(), run.size()))
            for (Job* j : run) errs[(size_t)(j - jobs.data())] = engine_error(g_engine);
    }
    return errs;
}

// ---------------------------------------------------------------------------
// generateDataUnit — pattern.go:94-116
// ---------------------------------------------------------------------------
std::pair<std::vector<uint8_t>, Error> generateDataUnit(int64_t unitCount, int64_t workerID,
                                                        const std::string& bucketName,
                                                        const object::Object* obj)
{
    // generateDataUnit itself does not truncate the bucket name; names that do
    // not format to 28 bytes would give a unit of another length, which the
    // engine does not produce
    Job job;
    job.is_fill = true;
    size_t unit_len = go_padded_len((const unsigned char*)bucketName.data(), bucketName.size(), 16) +
                      go_padded_len((const unsigned char*)obj->Key.data(), obj->Key.size(), 12) + 228;
    int rc_desc = OVAL_E_FIELD;
    if (unit_len == (size_t)dataUnitSize)
        rc_desc = make_desc(bucketName, obj, workerID, now_unix_micro(), (uint32_t)unitCount, 1, &job.desc, &unit_len);
    if (rc_desc || unit_len != (size_t)dataUnitSize) {
        char buf[160];
        std::snprintf(buf, sizeof buf,
                      "bucket and key do not format to 28 bytes: the data unit would be %zu bytes, not %d",
                      unit_len, dataUnitSize);
        return {{}, Error::New(buf)};
    }
    std::vector<uint8_t> unit(dataUnitSize);
    job.out = unit.data();
    execute(job);
    if (job.failed) return {{}, engine_error(g_engine)};
    return {std::move(unit), Error::Nil()};
}

// ---------------------------------------------------------------------------
// validDataUnit — pattern.go:136-180
// ---------------------------------------------------------------------------
namespace {

// Descriptor for validating `nUnits` units starting at unitCount `firstUnit`.
// An expected name longer than its field can never equal the trimmed field
// (pattern.go:140,147): that name is wrong whatever the unit holds; the other
// fields are still checked on the device.  Returns the mask of such names.
uint32_t validation_desc(const std::string& expectedBucketName, const object::Object* obj, int64_t workerID,
                         uint32_t firstUnit, uint32_t nUnits, oval_obj_desc* d, bool* ok)
{
    uint32_t forced = 0;
    std::string b = expectedBucketName;
    object::Object o = *obj;
    if (b.size() > (size_t)object::MaxBucketNameLength) { b.clear(); forced |= OVAL_BAD_BUCKET; }
    if (o.Key.size() > (size_t)object::MaxKeyLength) { o.Key.clear(); forced |= OVAL_BAD_KEY; }
    size_t unit_len = 0;
    *ok = make_desc(b, &o, workerID, 0, firstUnit, nUnits, d, &unit_len) == OVAL_OK;
    return forced;
}

}  // namespace

Error validDataUnit(int64_t unitCount, int64_t workerID, const std::string& expectedBucketName,
                    const object::Object* obj, const uint8_t* data)
{
    Job job;
    job.is_fill = false;
    job.in = data;
    bool ok = false;
    const uint32_t forced = validation_desc(expectedBucketName, obj, workerID, (uint32_t)unitCount, 1, &job.desc, &ok);
    if (!ok) return Error::New(std::string("oval engine failure: ") + oval_last_error());
    execute(job);
    if (job.failed) return engine_error(g_engine);
    uint32_t mask = (job.res.first_bad_unit == OVAL_NO_BAD_UNIT) ? 0u : job.res.first_bad_fields;
    mask |= forced;
    if (mask == 0) return Error::Nil();
    return Error::New(unit_error_text(mask, unitCount, workerID, expectedBucketName, obj, data));
}

// ---------------------------------------------------------------------------
// Valid — pattern.go:118-134: read the body, device pass over the complete
// units, then the reference's precedence (a bad unit wins over a short read).
// ---------------------------------------------------------------------------
namespace {

struct ValidState {
    std::string bucket16;
    int64_t nUnits = 0;
    size_t avail = 0, complete = 0;
    uint32_t forced = 0;
    bool has_job = false;
};

Error prepare_valid(int64_t workerID, const std::string& expectedBucketName, const object::Object* obj,
                    const uint8_t* body, size_t avail, ValidState& st, Job& job)
{
    st.bucket16 = truncate_bucket(expectedBucketName);
    st.nUnits = obj->Size / dataUnitSize;
    st.has_job = false;
    if (st.nUnits <= 0) return Error::Nil();                 // the loop at :123 does not run
    if (st.nUnits > 0xFFFFFFF0ll) return Error::New("object size beyond the engine's 1 TiB limit");
    const size_t want = (size_t)st.nUnits * dataUnitSize;
    st.avail = avail < want ? avail : want;
    st.complete = st.avail / dataUnitSize;
    if (st.complete == 0) return Error::Nil();
    job.is_fill = false;
    job.in = body;
    bool ok = false;
    st.forced = validation_desc(st.bucket16, obj, workerID, 0, (uint32_t)st.complete, &job.desc, &ok);
    if (!ok) return Error::New(std::string("oval engine failure: ") + oval_last_error());
    st.has_job = true;
    return Error::Nil();
}

Error finish_valid(int64_t workerID, const object::Object* obj, const uint8_t* body, const ValidState& st,
                   const Job& job)
{
    if (st.nUnits <= 0) return Error::Nil();
    if (st.has_job) {
        if (job.failed) return engine_error(g_engine);
        if (st.forced) {
            // a name longer than its field fails in the very first unit; the
            // device still says which of unit 0's other fields are wrong
            uint32_t mask = st.forced;
            if (job.res.first_bad_unit == 0) mask |= job.res.first_bad_fields;
            return Error::New(unit_error_text(mask, 0, workerID, st.bucket16, obj, body));
        }
        if (job.res.first_bad_unit != OVAL_NO_BAD_UNIT) {
            const uint8_t* unit = body + (size_t)job.res.first_bad_unit * dataUnitSize;
            return Error::New(unit_error_text(job.res.first_bad_fields, job.res.first_bad_unit, workerID,
                                              st.bucket16, obj, unit));
        }
    }
    if (st.complete < (size_t)st.nUnits) {
        // :125-127 the unit that could not be read in full
        const size_t n = st.avail - st.complete * dataUnitSize;
        char head[96];
        std::snprintf(head, sizeof head, "could not read some data. (expected: %dbyte, actual: %zubyte)\n",
                      dataUnitSize, n);
        return Error::New(std::string(head) + dump(hexDump(body + st.complete * dataUnitSize, n)));
    }
    return Error::Nil();
}

}  // namespace

Error ValidBytes(int64_t workerID, const std::string& expectedBucketName, const object::Object* obj,
                 const uint8_t* body, size_t bodyLen)
{
    ValidState st;
    Job job;
    Error err = prepare_valid(workerID, expectedBucketName, obj, body, bodyLen, st, job);
    if (!err.nil()) return err;
    if (st.has_job) execute(job);
    return finish_valid(workerID, obj, body, st, job);
}

namespace {

// Validates units [firstUnit, firstUnit + n) of an object, held in `data`;
// returns the reference's error for the first bad one, or nil.
Error validate_span(int64_t workerID, const std::string& bucket16, const object::Object* obj,
                    uint32_t firstUnit, const uint8_t* data, uint32_t n)
{
    Job job;
    job.is_fill = false;
    job.in = data;
    bool ok = false;
    const uint32_t forced = validation_desc(bucket16, obj, workerID, firstUnit, n, &job.desc, &ok);
    if (!ok) return Error::New(std::string("oval engine failure: ") + oval_last_error());
    execute(job);
    if (job.failed) return engine_error(g_engine);
    if (forced) {
        // a name longer than its field fails in the span's very first unit
        uint32_t mask = forced;
        if (job.res.first_bad_unit == firstUnit) mask |= job.res.first_bad_fields;
        return Error::New(unit_error_text(mask, firstUnit, workerID, bucket16, obj, data));
    }
    if (job.res.first_bad_unit != OVAL_NO_BAD_UNIT) {
        const uint8_t* unit = data + (size_t)(job.res.first_bad_unit - firstUnit) * dataUnitSize;
        return Error::New(unit_error_text(job.res.first_bad_fields, job.res.first_bad_unit, workerID, bucket16,
                                          obj, unit));
    }
    return Error::Nil();
}

size_t g_stream_chunk_bytes = 32u << 20;

}  // namespace

void SetStreamChunkBytes(size_t bytes)
{
    if (bytes < (size_t)dataUnitSize) bytes = dataUnitSize;
    g_stream_chunk_bytes = bytes - bytes % dataUnitSize;
}

// The body is consumed in chunks of at most SetStreamChunkBytes() (32 MiB by
// default) so that a multipart-sized object needs neither a body-sized host
// buffer nor a body-sized device buffer.  Reading stops at the first chunk
// that holds a bad unit, as the reference stops at the first bad unit.
Error Valid(int64_t workerID, const std::string& expectedBucketName, const object::Object* obj,
            Reader& reader)
{
    const std::string bucket16 = truncate_bucket(expectedBucketName);
    const int64_t nUnits = obj->Size / dataUnitSize;
    if (nUnits <= 0) return Error::Nil();                    // the loop at :123 does not run
    if (nUnits > 0xFFFFFFF0ll) return Error::New("object size beyond the engine's 1 TiB limit");
    const size_t chunk_units = g_stream_chunk_bytes / dataUnitSize;
    std::vector<uint8_t> buf((size_t)std::min<int64_t>(nUnits, (int64_t)chunk_units) * dataUnitSize);
    int64_t done = 0;
    while (done < nUnits) {
        const size_t want = (size_t)std::min<int64_t>(nUnits - done, (int64_t)chunk_units) * dataUnitSize;
        // io.ReadFull unit by unit (:124) == read until `want` bytes or EOF
        size_t got = 0;
        while (got < want) {
            const size_t n = reader.Read(buf.data() + got, want - got);
            if (n == 0) break;
            got += n;
        }
        const size_t complete = got / dataUnitSize;
        if (complete > 0) {
            Error err = validate_span(workerID, bucket16, obj, (uint32_t)done, buf.data(), (uint32_t)complete);
            if (!err.nil()) return err;                      // :128-131 a bad unit wins over a later short read
        }
        if (got < want) {
            // :125-127 the unit that could not be read in full
            const size_t n = got - complete * dataUnitSize;
            char head[96];
            std::snprintf(head, sizeof head, "could not read some data. (expected: %dbyte, actual: %zubyte)\n",
                          dataUnitSize, n);
            return Error::New(std::string(head) + dump(hexDump(buf.data() + complete * dataUnitSize, n)));
        }
        done += (int64_t)complete;
    }
    return Error::Nil();
}

std::vector<Error> ValidBatch(const std::vector<ValidItem>& items)
{
    std::vector<Error> errs(items.size());
    std::vector<Job> jobs(items.size());
    std::vector<ValidState> sts(items.size());
    std::vector<Job*> run;
    for (size_t i = 0; i < items.size(); ++i) {
        const ValidItem& it = items[i];
        errs[i] = prepare_valid(it.workerID, it.expectedBucketName, it.obj, it.body, it.bodyLen, sts[i], jobs[i]);
        if (errs[i].nil() && sts[i].has_job) run.push_back(&jobs[i]);
    }
    if (!run.empty()) {
        std::lock_guard<std::mutex> ek(g_engine.mu);
        run_jobs_locked(g_engine, run.data(), run.size());
    }
    for (size_t i = 0; i < items.size(); ++i)
        if (errs[i].nil())
            errs[i] = finish_valid(items[i].workerID, items[i].obj, items[i].body, sts[i], jobs[i]);
    return errs;
}

// ---------------------------------------------------------------------------
// hex.Dump and dump — pattern.go:182-206
// ---------------------------------------------------------------------------
std::string hexDump(const uint8_t* data, size_t n)
{
    static const char hexd[] = "0123456789abcdef";
    std::string out;
    out.reserve((n / 16 + 1) * 79);
    char right[18];
    size_t used = 0;
    for (size_t i = 0; i < n; ++i) {
        if (used == 0) {
            const uint32_t off = (uint32_t)i;
            for (int sh = 28; sh >= 0; sh -= 4) out += hexd[(off >> sh) & 15];
            out += "  ";
        }
        out += hexd[data[i] >> 4];
        out += hexd[data[i] & 15];
        out += ' ';
        if (used == 7) out += ' ';
        else if (used == 15) out += " |";
        right[used] = (data[i] < 32 || data[i] > 126) ? '.' : (char)data[i];
        if (++used == 16) {
            out.append(right, 16);
            out += "|\n";
            used = 0;
        }
    }
    if (used != 0) {
        const size_t nbytes = used;
        for (; used < 16; ++used) {
            out += "   ";
            if (used == 7) out += ' ';
            else if (used == 15) out += " |";
        }
        out.append(right, nbytes);
        out += "|\n";
    }
    return out;
}

std::string dump(const std::string& data)
{
    if (data.empty()) return "";
    static const char* const byteExplanation[3] = {
        "          ^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^ bucket name\n",
        "          ^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^ key name\n"
        "                                               ^^^^^^^^^^^ write count\n",
        "          ^^^^^^^^^^^ byte offset in this object\n"
        "                      ^^^^^^^^^^^ worker ID\n"
        "                                   ^^^^^^^^^^^^^^^^^^^^^^^ unix time (micro sec)\n",
    };
    const size_t lineSize = 79;
    std::string output;
    size_t pos = 0;
    for (const char* exp : byteExplanation) {
        if (data.size() - pos < lineSize) {
            output.append(data, pos, std::string::npos);
            output += exp;
            return output;
        }
        output.append(data, pos, lineSize);
        output += exp;
        pos += lineSize;
    }
    output.append(data, pos, std::string::npos);
    return output;
}

}  // namespace pattern
}  // namespace oval

// ---------------------------------------------------------------------------
// C entry points (include/oval_pattern.h)
// ---------------------------------------------------------------------------
namespace {

int put_error(const oval::Error& e, char* err, size_t cap)
{
    if (err && cap) {
        const size_t n = e.msg.size() < cap - 1 ? e.msg.size() : cap - 1;
        std::memcpy(err, e.msg.data(), n);
        err[n] = 0;
    }
    if (e.nil()) return 0;
    return e.msg.rfind("oval engine failure: ", 0) == 0 ? OVAL_E_HIP : 1;
}

size_t put_text(const std::string& s, char* out, size_t cap)
{
    if (out && cap) {
        const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        std::memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return s.size();
}

oval::object::Object make_obj(const char* key, size_t klen, int64_t size, int64_t wc)
{
    oval::object::Object o;
    o.Key.assign(key ? key : "", klen);
    o.Size = size;
    o.WriteCount = wc;
    return o;
}

}  // namespace

extern "C" {

int oval_pattern_init(int device_id) { return oval::pattern::Init(device_id); }
void oval_pattern_shutdown(void) { oval::pattern::Shutdown(); }
void oval_pattern_set_clock(int64_t unix_micro) { oval::pattern::SetClock(unix_micro); }
void oval_pattern_seed(uint64_t seed) { oval::pattern::Seed(seed); }

int oval_pattern_decide_size(int64_t min_size, int64_t max_size, int64_t* size_out, char* err, size_t err_cap)
{
    auto r = oval::pattern::DecideSize(min_size, max_size);
    if (size_out) *size_out = r.first;
    return put_error(r.second, err, err_cap);
}

int oval_pattern_decide_size_y(int64_t min_size, int64_t max_size, double y, int64_t* size_out, char* err,
                               size_t err_cap)
{
    auto r = oval::pattern::DecideSizeFrom(min_size, max_size, y);
    if (size_out) *size_out = r.first;
    return put_error(r.second, err, err_cap);
}

int oval_pattern_generate(int64_t data_size, int64_t worker_id, const char* bucket, size_t bucket_len,
                          const char* key, size_t key_len, int64_t write_count, uint8_t* out, char* err,
                          size_t err_cap)
{
    const oval::object::Object obj = make_obj(key, key_len, 0, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    return put_error(oval::pattern::GenerateInto(data_size, worker_id, b, &obj, out), err, err_cap);
}

int oval_pattern_generate_data_unit(int64_t unit_count, int64_t worker_id, const char* bucket,
                                    size_t bucket_len, const char* key, size_t key_len, int64_t write_count,
                                    uint8_t* out, char* err, size_t err_cap)
{
    const oval::object::Object obj = make_obj(key, key_len, 0, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    auto r = oval::pattern::generateDataUnit(unit_count, worker_id, b, &obj);
    if (r.second.nil() && out) std::memcpy(out, r.first.data(), r.first.size());
    return put_error(r.second, err, err_cap);
}

int oval_pattern_valid(int64_t worker_id, const char* bucket, size_t bucket_len, const char* key, size_t key_len,
                       int64_t size, int64_t write_count, const uint8_t* body, size_t body_len, char* err,
                       size_t err_cap)
{
    const oval::object::Object obj = make_obj(key, key_len, size, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    return put_error(oval::pattern::ValidBytes(worker_id, b, &obj, body, body_len), err, err_cap);
}

int oval_pattern_valid_data_unit(int64_t unit_count, int64_t worker_id, const char* bucket, size_t bucket_len,
                                 const char* key, size_t key_len, int64_t write_count, const uint8_t* data,
                                 char* err, size_t err_cap)
{
    const oval::object::Object obj = make_obj(key, key_len, 0, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    return put_error(oval::pattern::validDataUnit(unit_count, worker_id, b, &obj, data), err, err_cap);
}

namespace {
class CallbackReader : public oval::pattern::Reader {
public:
    CallbackReader(oval_read_fn fn, void* user) : fn_(fn), user_(user) {}
    size_t Read(uint8_t* p, size_t n) override { return fn_ ? fn_(user_, p, n) : 0; }
private:
    oval_read_fn fn_;
    void* user_;
};
}  // namespace

int oval_pattern_valid_stream(int64_t worker_id, const char* bucket, size_t bucket_len, const char* key,
                              size_t key_len, int64_t size, int64_t write_count, oval_read_fn read, void* user,
                              char* err, size_t err_cap)
{
    const oval::object::Object obj = make_obj(key, key_len, size, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    CallbackReader rd(read, user);
    return put_error(oval::pattern::Valid(worker_id, b, &obj, rd), err, err_cap);
}

void oval_pattern_set_stream_chunk(size_t bytes) { oval::pattern::SetStreamChunkBytes(bytes); }

void oval_pattern_set_batching(int enable, int window_us, size_t max_bytes)
{
    oval::pattern::SetMicroBatching(enable != 0, window_us, max_bytes);
}

void oval_pattern_batching_stats(uint64_t* batches, uint64_t* jobs)
{
    oval::pattern::MicroBatchingStats(batches, jobs);
}

int oval_pattern_generate_batch(size_t n, const int64_t* data_sizes, const int64_t* worker_ids,
                                const char* const* buckets, const size_t* bucket_lens, const char* const* keys,
                                const size_t* key_lens, const int64_t* write_counts, uint8_t* const* outs,
                                int* rcs, char* errs, size_t err_stride)
{
    std::vector<oval::object::Object> objs(n);
    std::vector<oval::pattern::GenerateItem> items(n);
    for (size_t i = 0; i < n; ++i) {
        objs[i] = make_obj(keys[i], key_lens[i], 0, write_counts[i]);
        items[i].dataSize = data_sizes[i];
        items[i].workerID = worker_ids[i];
        items[i].bucketName.assign(buckets[i] ? buckets[i] : "", bucket_lens[i]);
        items[i].obj = &objs[i];
        items[i].out = outs[i];
    }
    const std::vector<oval::Error> r = oval::pattern::GenerateBatch(items);
    int worst = 0;
    for (size_t i = 0; i < n; ++i) {
        const int rc = put_error(r[i], errs ? errs + i * err_stride : nullptr, errs ? err_stride : 0);
        if (rcs) rcs[i] = rc;
        if (rc < worst) worst = rc;
    }
    return worst;
}

int oval_pattern_valid_batch(size_t n, const int64_t* worker_ids, const char* const* buckets,
                             const size_t* bucket_lens, const char* const* keys, const size_t* key_lens,
                             const int64_t* sizes, const int64_t* write_counts, const uint8_t* const* bodies,
                             const size_t* body_lens, int* rcs, char* errs, size_t err_stride)
{
    std::vector<oval::object::Object> objs(n);
    std::vector<oval::pattern::ValidItem> items(n);
    for (size_t i = 0; i < n; ++i) {
        objs[i] = make_obj(keys[i], key_lens[i], sizes[i], write_counts[i]);
        items[i].workerID = worker_ids[i];
        items[i].expectedBucketName.assign(buckets[i] ? buckets[i] : "", bucket_lens[i]);
        items[i].obj = &objs[i];
        items[i].body = bodies[i];
        items[i].bodyLen = body_lens[i];
    }
    const std::vector<oval::Error> r = oval::pattern::ValidBatch(items);
    int worst = 0;
    for (size_t i = 0; i < n; ++i) {
        const int rc = put_error(r[i], errs ? errs + i * err_stride : nullptr, errs ? err_stride : 0);
        if (rcs) rcs[i] = rc;
        if (rc < worst) worst = rc;
    }
    return worst;
}

size_t oval_pattern_dump(const char* text, size_t n, char* out, size_t cap)
{
    return put_text(oval::pattern::dump(std::string(text ? text : "", n)), out, cap);
}

size_t oval_pattern_hex_dump(const uint8_t* data, size_t n, char* out, size_t cap)
{
    return put_text(oval::pattern::hexDump(data, n), out, cap);
}

size_t oval_object_generate_key(int64_t obj_id, char* out, size_t cap)
{
    return put_text(oval::object::generateKey(obj_id), out, cap);
}

}  // extern "C"
