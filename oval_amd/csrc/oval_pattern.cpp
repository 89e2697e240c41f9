// oval_pattern.cpp — host-side mirror of internal/pattern and internal/object
// (see oval_pattern.hpp).  Every byte Generate returns is written by the fill
// kernel and every verdict Valid returns is taken by the validate kernel; the
// host only formats names, moves bytes and words the reference's error text.
#include "oval_pattern.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <random>

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
// engine: one context, one reusable layout, growable staging buffers
// ---------------------------------------------------------------------------
namespace {

struct Engine {
    std::mutex mu;
    oval_ctx* ctx = nullptr;
    oval_layout* layout = nullptr;
    void* dev_body = nullptr;
    size_t dev_cap = 0;
    uint8_t* pinned = nullptr;
    size_t pinned_cap = 0;
    oval_obj_result* res_dev = nullptr;
    oval_obj_result* res_host = nullptr;   // pinned
    int64_t clock = -1;
    uint64_t rng[4] = {0, 0, 0, 0};
    bool seeded = false;
    std::string err;
};

Engine g_engine;

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
    rc = oval_layout_create(e.ctx, nullptr, 0, &e.layout);
    if (!rc) rc = oval_dev_alloc(e.ctx, sizeof(oval_obj_result), (void**)&e.res_dev);
    if (!rc) rc = oval_pinned_alloc(e.ctx, sizeof(oval_obj_result), (void**)&e.res_host);
    if (rc) {
        engine_fail(e, rc);
        if (e.res_dev) oval_dev_free(e.ctx, e.res_dev);
        if (e.layout) oval_layout_destroy(e.ctx, e.layout);
        oval_ctx_destroy(e.ctx);
        e.ctx = nullptr; e.layout = nullptr; e.res_dev = nullptr; e.res_host = nullptr;
        return rc;
    }
    return OVAL_OK;
}

int ensure_buffers_locked(Engine& e, size_t bytes, bool want_pinned)
{
    if (bytes > e.dev_cap) {
        if (e.dev_body) { oval_ctx_sync(e.ctx); oval_dev_free(e.ctx, e.dev_body); e.dev_body = nullptr; e.dev_cap = 0; }
        size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
        cap = (cap + 65535) & ~(size_t)65535;
        int rc = oval_dev_alloc(e.ctx, cap, &e.dev_body);
        if (rc) return engine_fail(e, rc);
        e.dev_cap = cap;
    }
    if (want_pinned && bytes > e.pinned_cap) {
        if (e.pinned) { oval_ctx_sync(e.ctx); oval_pinned_free(e.ctx, e.pinned); e.pinned = nullptr; e.pinned_cap = 0; }
        size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
        int rc = oval_pinned_alloc(e.ctx, cap, (void**)&e.pinned);
        if (rc) return engine_fail(e, rc);
        e.pinned_cap = cap;
    }
    return OVAL_OK;
}

Error engine_error(const Engine& e)
{
    return Error::New("oval engine failure: " + e.err);
}

int64_t now_unix_micro(const Engine& e)
{
    if (e.clock >= 0) return e.clock;
    using namespace std::chrono;
    return duration_cast<microseconds>(system_clock::now().time_since_epoch()).count();
}

std::string truncate_bucket(const std::string& b)
{
    // pattern.go:74-76 and :119-121
    return b.size() > (size_t)object::MaxBucketNameLength ? b.substr(0, object::MaxBucketNameLength) : b;
}

// Builds the one descriptor of a single-object call.  Returns OVAL_E_FIELD (and
// the formatted unit length) when the names do not format to 28 bytes.
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

// Device fill of one descriptor into `out` (host memory).
Error fill_one_locked(Engine& e, const oval_obj_desc& d, uint8_t* out)
{
    const size_t bytes = (size_t)d.n_units * dataUnitSize;
    if (bytes == 0) return Error::Nil();
    int rc = ensure_buffers_locked(e, bytes, false);
    if (!rc) { rc = oval_layout_reset(e.ctx, e.layout, &d, 1); if (rc) engine_fail(e, rc); }
    if (!rc) { rc = oval_fill(e.ctx, e.layout, e.dev_body, e.dev_cap); if (rc) engine_fail(e, rc); }
    if (!rc) { rc = oval_copy_d2h(e.ctx, out, e.dev_body, bytes); if (rc) engine_fail(e, rc); }
    if (!rc) { rc = oval_ctx_sync(e.ctx); if (rc) engine_fail(e, rc); }
    return rc ? engine_error(e) : Error::Nil();
}

// Device validate of `n_units` units held in the pinned buffer.
Error validate_one_locked(Engine& e, const oval_obj_desc& d, oval_obj_result* res)
{
    const size_t bytes = (size_t)d.n_units * dataUnitSize;
    int rc = oval_layout_reset(e.ctx, e.layout, &d, 1);
    if (rc) { engine_fail(e, rc); return engine_error(e); }
    rc = oval_copy_h2d(e.ctx, e.dev_body, e.pinned, bytes);
    if (!rc) rc = oval_validate(e.ctx, e.layout, e.dev_body, e.dev_cap, e.res_dev, nullptr);
    if (!rc) rc = oval_copy_d2h(e.ctx, e.res_host, e.res_dev, sizeof(oval_obj_result));
    if (!rc) rc = oval_ctx_sync(e.ctx);
    if (rc) { engine_fail(e, rc); return engine_error(e); }
    *res = *e.res_host;
    return Error::Nil();
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

void Shutdown()
{
    Engine& e = g_engine;
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
    e.res_dev = nullptr; e.res_host = nullptr; e.dev_cap = 0; e.pinned_cap = 0;
}

void SetClock(int64_t unixMicro)
{
    std::lock_guard<std::mutex> lk(g_engine.mu);
    g_engine.clock = unixMicro;
}

void Seed(uint64_t seed)
{
    std::lock_guard<std::mutex> lk(g_engine.mu);
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
        std::lock_guard<std::mutex> lk(g_engine.mu);
        y = float64_locked(g_engine);
    }
    return DecideSizeFrom(minSize, maxSize, y);
}

// ---------------------------------------------------------------------------
// Generate — pattern.go:73-92
// ---------------------------------------------------------------------------
Error GenerateInto(int64_t dataSize, int64_t workerID, const std::string& bucketName,
                   const object::Object* obj, uint8_t* out)
{
    const std::string bucket16 = truncate_bucket(bucketName);
    if (dataSize < 0) return Error::New("negative data size");   // make([]byte, 0, dataSize) panics in Go
    const int64_t nUnits = dataSize / dataUnitSize;
    if (nUnits > 0xFFFFFFF0ll) return Error::New("data size beyond the engine's 1 TiB limit");

    Engine& e = g_engine;
    std::lock_guard<std::mutex> lk(e.mu);
    oval_obj_desc d;
    size_t unit_len = 0;
    const int rc_desc = make_desc(bucket16, obj, workerID, now_unix_micro(e), 0, (uint32_t)nUnits, &d, &unit_len);
    // pattern.go:87-89: the length check that catches over-long fields and
    // sizes that are not a multiple of 256
    const int64_t produced = nUnits * (int64_t)unit_len;
    if (produced != dataSize) {
        char buf[128];
        std::snprintf(buf, sizeof buf, "generated data size is wrong. (expected: %lld, actual: %lld)",
                      (long long)dataSize, (long long)produced);
        return Error::New(buf);
    }
    if (nUnits == 0) return Error::Nil();
    if (rc_desc) { e.err = oval_last_error(); return engine_error(e); }
    if (int rc = ensure_engine_locked(e, 0)) { (void)rc; return engine_error(e); }
    return fill_one_locked(e, d, out);
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

// ---------------------------------------------------------------------------
// generateDataUnit — pattern.go:94-116
// ---------------------------------------------------------------------------
std::pair<std::vector<uint8_t>, Error> generateDataUnit(int64_t unitCount, int64_t workerID,
                                                        const std::string& bucketName,
                                                        const object::Object* obj)
{
    Engine& e = g_engine;
    std::lock_guard<std::mutex> lk(e.mu);
    oval_obj_desc d;
    size_t unit_len = 0;
    // generateDataUnit itself does not truncate the bucket name; names that do
    // not format to 28 bytes would give a unit of another length, which the
    // engine does not produce
    const int rc_desc = bucketName.size() > (size_t)object::MaxBucketNameLength
                            ? OVAL_E_FIELD
                            : make_desc(bucketName, obj, workerID, now_unix_micro(e), (uint32_t)unitCount, 1, &d, &unit_len);
    if (bucketName.size() > (size_t)object::MaxBucketNameLength)
        unit_len = go_padded_len((const unsigned char*)bucketName.data(), bucketName.size(), 16) +
                   go_padded_len((const unsigned char*)obj->Key.data(), obj->Key.size(), 12) + 228;
    if (rc_desc || unit_len != (size_t)dataUnitSize) {
        char buf[160];
        std::snprintf(buf, sizeof buf,
                      "bucket and key do not format to 28 bytes: the data unit would be %zu bytes, not %d",
                      unit_len, dataUnitSize);
        return {{}, Error::New(buf)};
    }
    if (ensure_engine_locked(e, 0)) return {{}, engine_error(e)};
    std::vector<uint8_t> unit(dataUnitSize);
    Error err = fill_one_locked(e, d, unit.data());
    if (!err.nil()) unit.clear();
    return {std::move(unit), err};
}

// ---------------------------------------------------------------------------
// validDataUnit — pattern.go:136-180
// ---------------------------------------------------------------------------
namespace {

// validDataUnit with the engine lock held.  `data` may be e.pinned itself.
Error valid_unit_locked(Engine& e, int64_t unitCount, int64_t workerID, const std::string& expectedBucketName,
                        const object::Object* obj, const uint8_t* data)
{
    if (ensure_engine_locked(e, 0)) return engine_error(e);
    if (ensure_buffers_locked(e, dataUnitSize, true)) return engine_error(e);
    if (data != e.pinned) std::memcpy(e.pinned, data, dataUnitSize);
    // An expected name longer than its field can never equal the trimmed field
    // (pattern.go:140,147): that name is wrong whatever the unit holds.  The
    // other fields are still checked on the device.
    uint32_t forced = 0;
    std::string b = expectedBucketName;
    object::Object o = *obj;
    if (b.size() > (size_t)object::MaxBucketNameLength) { b.clear(); forced |= OVAL_BAD_BUCKET; }
    if (o.Key.size() > (size_t)object::MaxKeyLength) { o.Key.clear(); forced |= OVAL_BAD_KEY; }
    oval_obj_desc d;
    size_t unit_len = 0;
    if (make_desc(b, &o, workerID, 0, (uint32_t)unitCount, 1, &d, &unit_len)) {
        e.err = oval_last_error();
        return engine_error(e);
    }
    oval_obj_result r;
    Error err = validate_one_locked(e, d, &r);
    if (!err.nil()) return err;
    uint32_t mask = (r.first_bad_unit == OVAL_NO_BAD_UNIT) ? 0u : r.first_bad_fields;
    mask |= forced;
    if (mask == 0) return Error::Nil();
    return Error::New(unit_error_text(mask, unitCount, workerID, expectedBucketName, obj, e.pinned));
}

}  // namespace

Error validDataUnit(int64_t unitCount, int64_t workerID, const std::string& expectedBucketName,
                    const object::Object* obj, const uint8_t* data)
{
    Engine& e = g_engine;
    std::lock_guard<std::mutex> lk(e.mu);
    return valid_unit_locked(e, unitCount, workerID, expectedBucketName, obj, data);
}

// ---------------------------------------------------------------------------
// Valid — pattern.go:118-134
// ---------------------------------------------------------------------------
Error Valid(int64_t workerID, const std::string& expectedBucketName, const object::Object* obj,
            Reader& reader)
{
    const std::string bucket16 = truncate_bucket(expectedBucketName);
    const int64_t nUnits = obj->Size / dataUnitSize;
    if (nUnits <= 0) return Error::Nil();                    // the loop at :123 does not run
    if (nUnits > 0xFFFFFFF0ll) return Error::New("object size beyond the engine's 1 TiB limit");
    const size_t want = (size_t)nUnits * dataUnitSize;

    Engine& e = g_engine;
    std::lock_guard<std::mutex> lk(e.mu);
    if (ensure_engine_locked(e, 0)) return engine_error(e);
    if (ensure_buffers_locked(e, want, true)) return engine_error(e);

    // io.ReadFull unit by unit (:124) == read until `want` bytes or EOF
    size_t avail = 0;
    while (avail < want) {
        const size_t n = reader.Read(e.pinned + avail, want - avail);
        if (n == 0) break;
        avail += n;
    }
    const size_t complete = avail / dataUnitSize;

    if (complete > 0) {
        // a key longer than its field fails in the very first unit (pattern.go:147)
        if (obj->Key.size() > (size_t)object::MaxKeyLength)
            return valid_unit_locked(e, 0, workerID, bucket16, obj, e.pinned);
        oval_obj_desc d;
        size_t unit_len = 0;
        if (make_desc(bucket16, obj, workerID, 0, 0, (uint32_t)complete, &d, &unit_len)) {
            e.err = oval_last_error();
            return engine_error(e);
        }
        oval_obj_result r;
        Error err = validate_one_locked(e, d, &r);
        if (!err.nil()) return err;
        if (r.first_bad_unit != OVAL_NO_BAD_UNIT) {
            const uint8_t* unit = e.pinned + (size_t)r.first_bad_unit * dataUnitSize;
            return Error::New(unit_error_text(r.first_bad_fields, r.first_bad_unit, workerID, bucket16, obj, unit));
        }
    }
    if (complete < (size_t)nUnits) {
        // :125-127 the unit that could not be read in full
        const size_t n = avail - complete * dataUnitSize;
        char head[96];
        std::snprintf(head, sizeof head, "could not read some data. (expected: %dbyte, actual: %zubyte)\n",
                      dataUnitSize, n);
        return Error::New(std::string(head) + dump(hexDump(e.pinned + complete * dataUnitSize, n)));
    }
    return Error::Nil();
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
    oval::pattern::BytesReader rd(body, body_len);
    return put_error(oval::pattern::Valid(worker_id, b, &obj, rd), err, err_cap);
}

int oval_pattern_valid_data_unit(int64_t unit_count, int64_t worker_id, const char* bucket, size_t bucket_len,
                                 const char* key, size_t key_len, int64_t write_count, const uint8_t* data,
                                 char* err, size_t err_cap)
{
    const oval::object::Object obj = make_obj(key, key_len, 0, write_count);
    const std::string b(bucket ? bucket : "", bucket_len);
    return put_error(oval::pattern::validDataUnit(unit_count, worker_id, b, &obj, data), err, err_cap);
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
