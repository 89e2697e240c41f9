// oval_pattern.cpp — host-side mirror of internal/pattern and internal/object
// (see oval_pattern.hpp).  Every byte Generate returns is written by the fill
// kernel and every verdict Valid returns is taken by the validate kernel; the
// host only formats names, moves bytes and words the reference's error text.
#include "oval_pattern.hpp"

#include <algorithm>
#include <atomic>
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
#include "oval_capi_internal.h"
#include "oval_engine.hpp"
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
// Host state of the mirror: the clock Generate stamps units with and the
// source of DecideSize's rand.Float64().  All device work goes through
// oval::engine (oval_engine.hpp): the scalar API hands it one job at a time
// (micro-batched with the calls of other threads while that is enabled),
// GenerateBatch / ValidBatch hand it many.
// ---------------------------------------------------------------------------
namespace {

using engine::Job;

// The source of rand.Float64() (pattern.go:63).  Go's global source is one
// locked generator; with 255 worker threads calling DecideSize that lock is a
// convoy, so every thread draws from a stream of its own, derived from the seed
// and the order in which threads first ask: the first thread after Seed(s) gets
// the stream a single-threaded program has always got.
struct HostState {
    std::mutex mu;                      // seed and stream numbering
    std::atomic<int64_t> clock{-1};
    uint64_t seed = 0;
    bool seeded = false;
    std::atomic<uint64_t> epoch{1};     // bumped by Seed(): threads re-derive their stream
    uint64_t next_stream = 0;
};

HostState g_state;

uint64_t splitmix64(uint64_t& x)
{
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct ThreadRng {
    uint64_t s[4] = {0, 0, 0, 0};
    uint64_t epoch = 0;
};

// xoshiro256** -> uniform double in [0,1) with 53 random bits, like rand.Float64()
double next_float64()
{
    thread_local ThreadRng r;
    const uint64_t epoch = g_state.epoch.load(std::memory_order_acquire);
    if (r.epoch != epoch) {
        std::lock_guard<std::mutex> lk(g_state.mu);
        if (!g_state.seeded) {
            std::random_device rd;
            g_state.seed = ((uint64_t)rd() << 32) ^ rd();
            g_state.seeded = true;
        }
        // stream k starts from the seed advanced by k odd constants: stream 0 is the seed itself
        uint64_t x = g_state.seed + 0xD1342543DE82EF95ull * g_state.next_stream++;
        for (auto& w : r.s) w = splitmix64(x);
        r.epoch = g_state.epoch.load(std::memory_order_relaxed);
    }
    auto rotl = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
    uint64_t* s = r.s;
    const uint64_t result = rotl(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return (double)(result >> 11) * 0x1.0p-53;
}

Error engine_error(const Job& job)
{
    return Error::New("oval engine failure: " + job.err);
}

using engine::execute;

int64_t now_unix_micro()
{
    const int64_t pinned = g_state.clock.load(std::memory_order_relaxed);
    if (pinned >= 0) return pinned;
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
int Init(int deviceID) { return engine::init(deviceID); }

int InitEngines(const int* devices, int n) { return engine::init_engines(devices, n); }

int EngineCount() { return engine::engine_count(); }

void SetCallerSpin(int nanos, int maxSpinners)
{
    engine::set_caller_spin(nanos);
    engine::set_spin_budget(maxSpinners);
}

void TestLoseFlags(int n) { engine::test_lose_flags(n); }

void DebugDump(FILE* out) { engine::debug_dump(out); }

int QuarantinedSlots() { return engine::quarantined_slots(); }

void SetMicroBatching(bool enable, int windowMicros, size_t maxBytes)
{
    engine::set_batching(enable, windowMicros, maxBytes);
}

void SetBatchLanes(int lanes, int blockingSync) { engine::set_lanes(lanes, blockingSync); }

void SetBatchShape(size_t chunkBytes, int depth) { engine::set_run_shape(chunkBytes, depth); }

void SetFlagCompletion(bool on) { engine::set_flag_completion(on); }

void SetDirectCallers(int callers) { engine::set_direct_max(callers); }

void SetWakeTuning(int fanout, int roots, int spinNanos)
{
    engine::set_wake_tree(fanout, roots);
    if (spinNanos >= 0) engine::set_caller_spin(spinNanos);
}

void MicroBatchingStats(uint64_t* batches, uint64_t* jobs) { engine::stats(batches, jobs); }

void LaneProfile(uint64_t out[8], bool reset) { engine::lane_profile(out, reset); }

void Shutdown() { engine::shutdown(); }

void SetClock(int64_t unixMicro) { g_state.clock.store(unixMicro, std::memory_order_relaxed); }

void Seed(uint64_t seed)
{
    std::lock_guard<std::mutex> lk(g_state.mu);
    g_state.seed = seed;
    g_state.seeded = true;
    g_state.next_stream = 0;
    g_state.epoch.fetch_add(1, std::memory_order_release);
}

std::string LastEngineError() { return engine::last_error(); }

// ---------------------------------------------------------------------------
// io
// ---------------------------------------------------------------------------
const uint8_t* BytesReader::Contiguous(size_t want, size_t* got)
{
    const size_t left = n_ - pos_;
    const uint8_t* p = p_ + pos_;
    *got = want < left ? want : left;
    pos_ += *got;
    return p;
}

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
    return DecideSizeFrom(minSize, maxSize, next_float64());
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
    return job.failed ? engine_error(job) : Error::Nil();
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

// fn(first, last) over the items of a batch form; a large batch's items are
// prepared (names padded, descriptors built) and judged on several host threads.
template <class Fn>
static void for_item_ranges(size_t n, const Fn& fn)
{
    constexpr size_t kPerRange = 2048;
    if (n < 2 * kPerRange) {
        fn((size_t)0, n);
        return;
    }
    const size_t parts = (n + kPerRange - 1) / kPerRange;
    oval_internal_parallel_for(parts, [&](size_t p) { fn(p * kPerRange, (p + 1) * kPerRange < n ? (p + 1) * kPerRange : n); });
}

std::vector<Error> GenerateBatch(const std::vector<GenerateItem>& items)
{
    std::vector<Error> errs(items.size());
    std::vector<Job> jobs(items.size());
    std::vector<Job*> run;
    for_item_ranges(items.size(), [&](size_t first, size_t last) {
        for (size_t i = first; i < last; ++i) {
            const GenerateItem& it = items[i];
            errs[i] = prepare_generate(it.dataSize, it.workerID, it.bucketName, it.obj, it.out, jobs[i]);
        }
    });
    for (size_t i = 0; i < items.size(); ++i)
        if (errs[i].nil() && jobs[i].desc.n_units > 0) run.push_back(&jobs[i]);
    if (!run.empty()) {
        engine::run(run.data(), run.size());
        for (Job* j : run)
            if (j->failed) errs[(size_t)(j - jobs.data())] = engine_error(*j);
    }
    return errs;
}

// ---------------------------------------------------------------------------
// GenerateParts — Generate (pattern.go:73-92) whose result is consumed in parts
// (internal/s3client/s3client.go:164-236)
// ---------------------------------------------------------------------------
std::pair<GeneratedParts*, Error> GenerateParts(int64_t dataSize, int64_t workerID, const std::string& bucketName,
                                                const object::Object* obj, int64_t partSize)
{
    if (partSize <= 0) return {nullptr, Error::New("oval engine failure: part size must be positive")};
    Job job;
    Error err = prepare_generate(dataSize, workerID, bucketName, obj, nullptr, job);
    if (!err.nil()) return {nullptr, err};
    std::string eng_err;
    engine::Parts* impl = engine::parts_open(job.desc, &eng_err);
    if (!impl) return {nullptr, Error::New("oval engine failure: " + eng_err)};
    GeneratedParts* p = new GeneratedParts;
    p->impl_ = impl;
    p->size_ = (int64_t)job.desc.n_units * dataUnitSize;
    p->part_size_ = partSize;
    p->count_ = (p->size_ + partSize - 1) / partSize;
    p->outstanding_ = p->count_;
    p->state_.assign((size_t)p->count_, 0);
    p->mu_ = new std::mutex;
    if (p->count_ == 0) p->Close();                          // nothing to hand out
    return {p, Error::Nil()};
}

GeneratedParts::~GeneratedParts()
{
    Close();
    delete static_cast<std::mutex*>(mu_);
}

Error GeneratedParts::Part(int64_t i, const uint8_t** data, size_t* len)
{
    if (i < 0 || i >= count_) return Error::New("oval engine failure: part index out of range");
    // held across the wait: a Close() from another thread must not free the buffer under it
    std::lock_guard<std::mutex> lk(*static_cast<std::mutex*>(mu_));
    if (!impl_ || state_[(size_t)i] == 2) return Error::New("oval engine failure: part already released");
    state_[(size_t)i] = 1;
    engine::Parts* impl = static_cast<engine::Parts*>(impl_);
    const size_t begin = (size_t)(i * part_size_);
    const size_t end = (size_t)std::min<int64_t>(size_, (i + 1) * part_size_);
    std::string eng_err;
    const uint8_t* base = engine::parts_wait(impl, begin, end, &eng_err);
    if (!base) return Error::New("oval engine failure: " + eng_err);
    *data = base + begin;
    *len = end - begin;
    return Error::Nil();
}

Error GeneratedParts::Release(int64_t i)
{
    if (i < 0 || i >= count_) return Error::New("oval engine failure: part index out of range");
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(*static_cast<std::mutex*>(mu_));
        if (!impl_ || state_[(size_t)i] == 2) return Error::Nil();   // releasing twice is harmless
        state_[(size_t)i] = 2;
        last = --outstanding_ == 0;
    }
    if (last) Close();
    return Error::Nil();
}

void GeneratedParts::Close()
{
    engine::Parts* impl = nullptr;
    if (mu_) {
        std::lock_guard<std::mutex> lk(*static_cast<std::mutex*>(mu_));
        impl = static_cast<engine::Parts*>(impl_);
        impl_ = nullptr;
    }
    if (impl) engine::parts_close(impl);
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
    if (job.failed) return {{}, engine_error(job)};
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
    if (job.failed) return engine_error(job);
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
        if (job.failed) return engine_error(job);
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

// ---------------------------------------------------------------------------
// Generate / Valid in two steps, for callers that have something better to do
// than sleep while the device works (a goroutine that yields to its scheduler)
// ---------------------------------------------------------------------------
struct Pending {
    bool is_generate = false, submitted = false, started = false, finished = false;
    Job job;
    engine::Ticket ticket;
    ValidState st;
    object::Object obj;
    int64_t workerID = 0;
    const uint8_t* body = nullptr;
    Error result = Error::Nil();
};

namespace {
// The device step of a pending call: through the micro-batcher without waiting when it can take the job,
// else here and now.
// (no room in the micro-batcher right now: the job stays where it is and the next Poll offers it again)
void pending_start(Pending* p)
{
    const int how = engine::submit(p->job, p->ticket);
    if (how == 0) return;
    p->started = true;
    if (how > 0) p->submitted = true;
    else execute(p->job);
}
}  // namespace

Pending* GenerateBegin(int64_t dataSize, int64_t workerID, const std::string& bucketName, const object::Object* obj,
                       uint8_t* out)
{
    Pending* p = new Pending;
    p->is_generate = true;
    p->result = prepare_generate(dataSize, workerID, bucketName, obj, out, p->job);
    if (!p->result.nil() || p->job.desc.n_units == 0) {
        p->finished = true;
        return p;
    }
    pending_start(p);
    return p;
}

Pending* ValidBegin(int64_t workerID, const std::string& expectedBucketName, const object::Object* obj,
                    const uint8_t* body, size_t bodyLen)
{
    Pending* p = new Pending;
    p->obj = *obj;
    p->workerID = workerID;
    p->body = body;
    p->result = prepare_valid(workerID, expectedBucketName, &p->obj, body, bodyLen, p->st, p->job);
    if (!p->result.nil() || !p->st.has_job) {
        if (p->result.nil()) p->result = finish_valid(workerID, &p->obj, body, p->st, p->job);
        p->finished = true;
        return p;
    }
    pending_start(p);
    return p;
}

bool Poll(Pending* p, Error* result)
{
    if (!p->finished) {
        if (!p->started) {
            pending_start(p);
            if (!p->started) return false;
        }
        if (p->submitted && !engine::try_complete(p->ticket, p->job)) return false;
        if (p->is_generate) p->result = p->job.failed ? engine_error(p->job) : Error::Nil();
        else p->result = finish_valid(p->workerID, &p->obj, p->body, p->st, p->job);
        p->finished = true;
    }
    if (result) *result = p->result;
    delete p;
    return true;
}

namespace {

std::atomic<size_t> g_stream_chunk_bytes{32u << 20};

Error short_read_error(const uint8_t* partial, size_t n)
{
    // pattern.go:125-127 the unit that could not be read in full
    char head[96];
    std::snprintf(head, sizeof head, "could not read some data. (expected: %dbyte, actual: %zubyte)\n",
                  dataUnitSize, n);
    return Error::New(std::string(head) + dump(hexDump(partial, n)));
}

// The reference's error for the outcome of units [firstUnit, firstUnit + n) held in `data`, or nil.
Error span_verdict(const oval_obj_result& res, uint32_t forced, int64_t workerID, const std::string& bucket16,
                   const object::Object* obj, uint32_t firstUnit, const uint8_t* data)
{
    if (forced) {
        // a name longer than its field fails in the span's very first unit
        uint32_t mask = forced;
        if (res.first_bad_unit == firstUnit) mask |= res.first_bad_fields;
        return Error::New(unit_error_text(mask, firstUnit, workerID, bucket16, obj, data));
    }
    if (res.first_bad_unit != OVAL_NO_BAD_UNIT) {
        const uint8_t* unit = data + (size_t)(res.first_bad_unit - firstUnit) * dataUnitSize;
        return Error::New(unit_error_text(res.first_bad_fields, res.first_bad_unit, workerID, bucket16, obj, unit));
    }
    return Error::Nil();
}

// io.ReadFull unit by unit (pattern.go:124) == read until `want` bytes or EOF
size_t read_full(Reader& reader, uint8_t* p, size_t want)
{
    size_t got = 0;
    while (got < want) {
        const size_t n = reader.Read(p + got, want - got);
        if (n == 0) break;
        got += n;
    }
    return got;
}

// A body larger than one chunk: chunk k+1 is read straight into one half of a
// pinned buffer while chunk k, in the other half, is copied to the device and
// validated.  A chunk's verdict is collected after the next chunk has been
// read, so the reader may be asked for one chunk more than the reference would
// have consumed; which error is returned is not affected: the lowest bad unit
// wins, and a bad unit wins over a short read (pattern.go:124-131).
Error valid_streaming(int64_t workerID, const std::string& bucket16, const object::Object* obj, Reader& reader,
                      int64_t nUnits, size_t chunk_bytes)
{
    std::string eng_err;
    engine::Stream* st = engine::stream_open(chunk_bytes, (uint32_t)(uint64_t)workerID, &eng_err);
    if (!st) return Error::New("oval engine failure: " + eng_err);
    struct Pending { bool active = false; uint32_t first = 0, forced = 0; };
    Pending pend[2];
    Error result = Error::Nil();
    bool done = false;
    // collects the verdict of the chunk in `half`; true when it ends the validation
    auto collect = [&](int half) {
        Pending& p = pend[half];
        if (!p.active) return false;
        p.active = false;
        oval_obj_result res;
        if (engine::stream_wait(st, half, &res)) {
            result = Error::New("oval engine failure: " + engine::stream_error(st));
            return true;
        }
        result = span_verdict(res, p.forced, workerID, bucket16, obj, p.first, engine::stream_buffer(st, half));
        return !result.nil();
    };
    const size_t chunk_units = chunk_bytes / dataUnitSize;
    int64_t first = 0;
    for (int k = 0; first < nUnits && !done; ++k) {
        const int half = k & 1;
        const size_t want = (size_t)std::min<int64_t>(nUnits - first, (int64_t)chunk_units) * dataUnitSize;
        uint8_t* buf = engine::stream_buffer(st, half);     // its previous chunk (k-2) was collected in round k-1
        const size_t got = read_full(reader, buf, want);
        const size_t complete = got / dataUnitSize;
        if (complete > 0) {
            oval_obj_desc d;
            bool ok = false;
            pend[half].forced = validation_desc(bucket16, obj, workerID, (uint32_t)first, (uint32_t)complete, &d, &ok);
            pend[half].first = (uint32_t)first;
            if (!ok) {
                result = Error::New(std::string("oval engine failure: ") + oval_last_error());
                break;
            }
            if (engine::stream_submit(st, half, d)) {
                result = Error::New("oval engine failure: " + engine::stream_error(st));
                break;
            }
            pend[half].active = true;
        }
        if (collect(half ^ 1)) break;                        // the chunk before this one
        if (got < want) {
            if (collect(half)) break;                        // a bad unit wins over the short read
            result = short_read_error(buf + complete * dataUnitSize, got - complete * dataUnitSize);
            done = true;
        }
        first += (int64_t)complete;
    }
    if (result.nil() && !done) {
        // the last one or two chunks are still on the device; the earlier one first
        const int last = pend[0].active && pend[1].active ? (pend[0].first < pend[1].first ? 0 : 1) : (pend[0].active ? 0 : 1);
        if (!collect(last)) collect(last ^ 1);
    }
    engine::stream_close(st);
    return result;
}

}  // namespace

void SetStreamChunkBytes(size_t bytes)
{
    if (bytes < (size_t)dataUnitSize) bytes = dataUnitSize;
    g_stream_chunk_bytes = bytes - bytes % dataUnitSize;
}

// A body of up to one chunk (SetStreamChunkBytes(), 32 MiB by default) is read
// whole and validated as one job, which the micro-batcher may fold into the
// same launch as other callers' jobs.  A larger body streams through a pinned
// double buffer so that neither a body-sized host buffer nor a body-sized
// device buffer is needed and the reads overlap the device work.
Error Valid(int64_t workerID, const std::string& expectedBucketName, const object::Object* obj,
            Reader& reader)
{
    const std::string bucket16 = truncate_bucket(expectedBucketName);
    const int64_t nUnits = obj->Size / dataUnitSize;
    if (nUnits <= 0) return Error::Nil();                    // the loop at :123 does not run
    if (nUnits > 0xFFFFFFF0ll) return Error::New("object size beyond the engine's 1 TiB limit");
    const size_t chunk_bytes = g_stream_chunk_bytes.load();
    const size_t want = (size_t)nUnits * dataUnitSize;
    size_t got = 0;
    const uint8_t* body = reader.Contiguous(want, &got);     // a reader over memory lends its bytes, whatever their number
    if (!body && want > chunk_bytes) return valid_streaming(workerID, bucket16, obj, reader, nUnits, chunk_bytes);
    if (!body) {
        thread_local std::vector<uint8_t> bounce;
        if (bounce.size() < want) bounce.resize(want);
        got = read_full(reader, bounce.data(), want);
        body = bounce.data();
    }
    return ValidBytes(workerID, expectedBucketName, obj, body, got);
}

std::vector<Error> ValidBatch(const std::vector<ValidItem>& items)
{
    std::vector<Error> errs(items.size());
    std::vector<Job> jobs(items.size());
    std::vector<ValidState> sts(items.size());
    std::vector<Job*> run;
    for_item_ranges(items.size(), [&](size_t first, size_t last) {
        for (size_t i = first; i < last; ++i) {
            const ValidItem& it = items[i];
            errs[i] = prepare_valid(it.workerID, it.expectedBucketName, it.obj, it.body, it.bodyLen, sts[i], jobs[i]);
        }
    });
    for (size_t i = 0; i < items.size(); ++i)
        if (errs[i].nil() && sts[i].has_job) run.push_back(&jobs[i]);
    if (!run.empty()) engine::run(run.data(), run.size());
    for_item_ranges(items.size(), [&](size_t first, size_t last) {
        for (size_t i = first; i < last; ++i)
            if (errs[i].nil())
                errs[i] = finish_valid(items[i].workerID, items[i].obj, items[i].body, sts[i], jobs[i]);
    });
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

// Handler of the batch forms: every item gets the status and the text of what was caught.
int batch_caught(size_t n, int* rcs, char* errs, size_t err_stride)
{
    const int rc = oval_internal_caught();
    for (size_t i = 0; i < n && (rcs || (errs && err_stride)); ++i) {
        if (rcs) rcs[i] = rc;
        if (errs && err_stride) {
            std::strncpy(errs + i * err_stride, oval_last_error(), err_stride - 1);
            errs[i * err_stride + err_stride - 1] = '\0';
        }
    }
    return rc;
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

int oval_pattern_init(int device_id)
try { return oval::pattern::Init(device_id); } catch (...) {
    return oval_internal_caught();
}
void oval_pattern_shutdown(void)
try { oval::pattern::Shutdown(); } catch (...) {
    (void)oval_internal_caught();
}
void oval_pattern_set_clock(int64_t unix_micro)
try { oval::pattern::SetClock(unix_micro); } catch (...) {
    (void)oval_internal_caught();
}
void oval_pattern_seed(uint64_t seed)
try { oval::pattern::Seed(seed); } catch (...) {
    (void)oval_internal_caught();
}

int oval_pattern_decide_size(int64_t min_size, int64_t max_size, int64_t* size_out, char* err, size_t err_cap)
try {
    auto r = oval::pattern::DecideSize(min_size, max_size);
    if (size_out) *size_out = r.first;
    return put_error(r.second, err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int oval_pattern_decide_size_y(int64_t min_size, int64_t max_size, double y, int64_t* size_out, char* err,
                               size_t err_cap)
try {
    auto r = oval::pattern::DecideSizeFrom(min_size, max_size, y);
    if (size_out) *size_out = r.first;
    return put_error(r.second, err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int oval_pattern_generate(int64_t data_size, int64_t worker_id, const char* bucket, size_t bucket_len,
                          const char* key, size_t key_len, int64_t write_count, uint8_t* out, char* err,
                          size_t err_cap)
try {
    const oval::object::Object obj = make_obj(key, key_len, 0, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    return put_error(oval::pattern::GenerateInto(data_size, worker_id, b, &obj, out), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int oval_pattern_generate_data_unit(int64_t unit_count, int64_t worker_id, const char* bucket,
                                    size_t bucket_len, const char* key, size_t key_len, int64_t write_count,
                                    uint8_t* out, char* err, size_t err_cap)
try {
    const oval::object::Object obj = make_obj(key, key_len, 0, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    auto r = oval::pattern::generateDataUnit(unit_count, worker_id, b, &obj);
    if (r.second.nil() && out) std::memcpy(out, r.first.data(), r.first.size());
    return put_error(r.second, err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int oval_pattern_valid(int64_t worker_id, const char* bucket, size_t bucket_len, const char* key, size_t key_len,
                       int64_t size, int64_t write_count, const uint8_t* body, size_t body_len, char* err,
                       size_t err_cap)
try {
    const oval::object::Object obj = make_obj(key, key_len, size, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    return put_error(oval::pattern::ValidBytes(worker_id, b, &obj, body, body_len), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int oval_pattern_valid_data_unit(int64_t unit_count, int64_t worker_id, const char* bucket, size_t bucket_len,
                                 const char* key, size_t key_len, int64_t write_count, const uint8_t* data,
                                 char* err, size_t err_cap)
try {
    const oval::object::Object obj = make_obj(key, key_len, 0, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    return put_error(oval::pattern::validDataUnit(unit_count, worker_id, b, &obj, data), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
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
try {
    const oval::object::Object obj = make_obj(key, key_len, size, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    CallbackReader rd(read, user);
    return put_error(oval::pattern::Valid(worker_id, b, &obj, rd), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

void oval_pattern_set_stream_chunk(size_t bytes)
try { oval::pattern::SetStreamChunkBytes(bytes); } catch (...) {
    (void)oval_internal_caught();
}

int oval_pattern_generate_parts(int64_t data_size, int64_t worker_id, const char* bucket, size_t bucket_len,
                                const char* key, size_t key_len, int64_t write_count, int64_t part_size,
                                oval_parts** out, char* err, size_t err_cap)
try {
    if (out) *out = nullptr;
    if (!out) return put_error(oval::Error::New("oval engine failure: out is NULL"), err, err_cap);
    const oval::object::Object obj = make_obj(key, key_len, 0, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    auto r = oval::pattern::GenerateParts(data_size, worker_id, b, &obj, part_size);
    *out = reinterpret_cast<oval_parts*>(r.first);
    return put_error(r.second, err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int64_t oval_parts_count(const oval_parts* parts)
try {
    return parts ? reinterpret_cast<const oval::pattern::GeneratedParts*>(parts)->Count() : 0;
} catch (...) {
    (void)oval_internal_caught();
    return -1;
}

int oval_parts_get(oval_parts* parts, int64_t index, const uint8_t** data, size_t* len, char* err, size_t err_cap)
try {
    if (!parts || !data || !len) return put_error(oval::Error::New("oval engine failure: NULL argument"), err, err_cap);
    return put_error(reinterpret_cast<oval::pattern::GeneratedParts*>(parts)->Part(index, data, len), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int oval_parts_release(oval_parts* parts, int64_t index)
try {
    if (!parts) return OVAL_E_INVALID;
    return reinterpret_cast<oval::pattern::GeneratedParts*>(parts)->Release(index).nil() ? 0 : OVAL_E_INVALID;
} catch (...) {
    return oval_internal_caught();
}

void oval_parts_close(oval_parts* parts)
try { delete reinterpret_cast<oval::pattern::GeneratedParts*>(parts); } catch (...) {
    (void)oval_internal_caught();
}

void oval_parts_pool_stats(size_t* buffers, size_t* bytes)
try { oval::engine::parts_pool_stats(buffers, bytes); } catch (...) {
    (void)oval_internal_caught();
}

void oval_pattern_set_batching(int enable, int window_us, size_t max_bytes)
try {
    oval::pattern::SetMicroBatching(enable != 0, window_us, max_bytes);
} catch (...) {
    (void)oval_internal_caught();
}

void oval_pattern_set_lanes(int lanes, int blocking_sync)
try { oval::pattern::SetBatchLanes(lanes, blocking_sync); } catch (...) {
    (void)oval_internal_caught();
}

void oval_pattern_set_direct_callers(int callers)
try { oval::pattern::SetDirectCallers(callers); } catch (...) {
    (void)oval_internal_caught();
}

int oval_pattern_generate_begin(int64_t data_size, int64_t worker_id, const char* bucket, size_t bucket_len,
                                const char* key, size_t key_len, int64_t write_count, uint8_t* out,
                                oval_pending** pending)
try {
    if (!pending) return oval_internal_fail(OVAL_E_INVALID, "pending is NULL");
    const oval::object::Object obj = make_obj(key, key_len, 0, write_count);
    *pending = reinterpret_cast<oval_pending*>(
        oval::pattern::GenerateBegin(data_size, worker_id, std::string(bucket ? bucket : "", bucket_len), &obj, out));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_pattern_valid_begin(int64_t worker_id, const char* bucket, size_t bucket_len, const char* key, size_t key_len,
                             int64_t size, int64_t write_count, const uint8_t* body, size_t body_len,
                             oval_pending** pending)
try {
    if (!pending) return oval_internal_fail(OVAL_E_INVALID, "pending is NULL");
    const oval::object::Object obj = make_obj(key, key_len, size, write_count);
    *pending = reinterpret_cast<oval_pending*>(
        oval::pattern::ValidBegin(worker_id, std::string(bucket ? bucket : "", bucket_len), &obj, body, body_len));
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_pattern_poll(oval_pending* pending, char* err, size_t err_cap)
try {
    if (!pending) return oval_internal_fail(OVAL_E_INVALID, "pending is NULL");
    oval::Error e = oval::Error::Nil();
    if (!oval::pattern::Poll(reinterpret_cast<oval::pattern::Pending*>(pending), &e)) return OVAL_PATTERN_PENDING;
    return put_error(e, err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}
int oval_pattern_init_engines(const int* devices, int n)
try { return oval::pattern::InitEngines(devices, n); } catch (...) {
    return oval_internal_caught();
}
int oval_pattern_engine_count(void)
try { return oval::pattern::EngineCount(); } catch (...) {
    return oval_internal_caught();
}
void oval_pattern_set_device_timeout(uint64_t ms)
try { (void)oval_set_device_timeout_ms(ms); } catch (...) {
    (void)oval_internal_caught();
}
void oval_pattern_set_caller_spin(int spin_ns, int max_spinners)
try { oval::pattern::SetCallerSpin(spin_ns, max_spinners); } catch (...) {
    (void)oval_internal_caught();
}
void oval_pattern_test_lose_flags(int n)
try { oval::pattern::TestLoseFlags(n); } catch (...) {
    (void)oval_internal_caught();
}
int oval_pattern_quarantined_slots(void)
try { return oval::pattern::QuarantinedSlots(); } catch (...) {
    return oval_internal_caught();
}
void oval_pattern_set_wake_tuning(int fanout, int roots, int spin_ns)
try { oval::pattern::SetWakeTuning(fanout, roots, spin_ns); } catch (...) {
    (void)oval_internal_caught();
}

void oval_pattern_batching_stats(uint64_t* batches, uint64_t* jobs)
try {
    oval::pattern::MicroBatchingStats(batches, jobs);
} catch (...) {
    (void)oval_internal_caught();
}

int oval_pattern_generate_batch(size_t n, const int64_t* data_sizes, const int64_t* worker_ids,
                                const char* const* buckets, const size_t* bucket_lens, const char* const* keys,
                                const size_t* key_lens, const int64_t* write_counts, uint8_t* const* outs,
                                int* rcs, char* errs, size_t err_stride)
try {
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
} catch (...) {
    return batch_caught(n, rcs, errs, err_stride);
}

int oval_pattern_valid_batch(size_t n, const int64_t* worker_ids, const char* const* buckets,
                             const size_t* bucket_lens, const char* const* keys, const size_t* key_lens,
                             const int64_t* sizes, const int64_t* write_counts, const uint8_t* const* bodies,
                             const size_t* body_lens, int* rcs, char* errs, size_t err_stride)
try {
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
} catch (...) {
    return batch_caught(n, rcs, errs, err_stride);
}

size_t oval_pattern_dump(const char* text, size_t n, char* out, size_t cap)
try {
    return put_text(oval::pattern::dump(std::string(text ? text : "", n)), out, cap);
} catch (...) {
    (void)oval_internal_caught();
    return 0;
}

size_t oval_pattern_hex_dump(const uint8_t* data, size_t n, char* out, size_t cap)
try {
    return put_text(oval::pattern::hexDump(data, n), out, cap);
} catch (...) {
    (void)oval_internal_caught();
    return 0;
}

size_t oval_object_generate_key(int64_t obj_id, char* out, size_t cap)
try {
    return put_text(oval::object::generateKey(obj_id), out, cap);
} catch (...) {
    (void)oval_internal_caught();
    return 0;
}

}  // extern "C"
