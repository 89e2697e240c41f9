// oval_pattern.hpp — host-side mirror of the reference's Go packages
// internal/object and internal/pattern, in front of the HIP engine.
//
// Names, argument meaning and error text follow the reference so that its
// callers (internal/runner/worker.go:64,77,85,109,140,200) and its tests
// (internal/pattern/pattern_test.go) read the same against this API.  A Go
// `error` becomes an oval::Error: `nil()` is Go's nil, otherwise `msg` is what
// err.Error() would return.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

namespace oval {

struct Error {
    bool ok = true;
    std::string msg;
    bool nil() const { return ok; }
    static Error Nil() { return Error{}; }
    static Error New(std::string m) { Error e; e.ok = false; e.msg = std::move(m); return e; }
};

namespace object {

// internal/object/object.go:13-19
constexpr int MaxBucketNameLength = 16;
constexpr int MaxKeyLength = 12;
constexpr const char* KeyShortPrefix = "ov";
constexpr int KeyPrefixLength = 6;

// internal/object/object.go:21-25
struct Object {
    std::string Key;
    int64_t Size = 0;
    int64_t WriteCount = 0;
    void Clear() { Size = 0; WriteCount = 0; }   // object.go:35-38
};

std::string generateKey(int64_t objID);          // object.go:48-50
Object NewObject(int64_t objID);                 // object.go:40-46

}  // namespace object

namespace pattern {

constexpr int dataUnitSize = 256;                          // pattern.go:18
constexpr int dataUnitHeaderSizeWithoutBucketAndKey = 20;  // pattern.go:19

// io.Reader: returns the number of bytes read into p (<= n); 0 means EOF.
class Reader {
public:
    virtual ~Reader() = default;
    virtual size_t Read(uint8_t* p, size_t n) = 0;
    // A reader over memory may lend its next min(want, remaining) bytes instead
    // of copying them (and advances past them); NULL = not supported, use Read.
    virtual const uint8_t* Contiguous(size_t want, size_t* got) { (void)want; (void)got; return nullptr; }
};

// bytes.NewReader
class BytesReader : public Reader {
public:
    BytesReader(const uint8_t* data, size_t len) : p_(data), n_(len) {}
    size_t Read(uint8_t* p, size_t n) override;
    const uint8_t* Contiguous(size_t want, size_t* got) override;
private:
    const uint8_t* p_;
    size_t n_;
    size_t pos_ = 0;
};

// pattern.go:22-71
std::pair<int64_t, Error> DecideSize(int64_t minSize, int64_t maxSize);
std::pair<int64_t, Error> DecideSizeFrom(int64_t minSize, int64_t maxSize, double y);

// pattern.go:73-92
std::pair<std::vector<uint8_t>, Error> Generate(int64_t dataSize, int64_t workerID,
                                                const std::string& bucketName,
                                                const object::Object* obj);
// the same into caller-provided memory (dataSize bytes)
Error GenerateInto(int64_t dataSize, int64_t workerID, const std::string& bucketName,
                   const object::Object* obj, uint8_t* out);

// Generate for a body that leaves in parts.  The reference's multipartUpload
// (internal/s3client/s3client.go:164-236) re-slices the []byte Generate returned
// into parts of multipartThresh bytes (body[:partSize] at :185, body =
// body[partSize:] at :204) and sends them one after the other.  Here the body is
// generated once into pinned memory owned by the library (no copy into caller
// memory) and arrives there slice by slice while earlier parts are already on
// their way out: Part(i) waits only for the bytes of part i.
class GeneratedParts {
public:
    ~GeneratedParts();
    int64_t Size() const { return size_; }
    int64_t PartSize() const { return part_size_; }
    int64_t Count() const { return count_; }          // ceil(Size / PartSize): the UploadPart calls of :176-211
    // body[i*PartSize : min((i+1)*PartSize, Size)]; valid until Release(i) or Close().
    Error Part(int64_t i, const uint8_t** data, size_t* len);
    // The caller is done with part i.  When every part has been released the
    // pinned buffer goes back to the library's pool.
    Error Release(int64_t i);
    // Gives up whatever is still held (the abort path, s3client.go:190-203).
    void Close();
private:
    friend std::pair<GeneratedParts*, Error> GenerateParts(int64_t, int64_t, const std::string&,
                                                            const object::Object*, int64_t);
    GeneratedParts() = default;
    void* impl_ = nullptr;
    int64_t size_ = 0, part_size_ = 0, count_ = 0, outstanding_ = 0;
    std::vector<uint8_t> state_;                       // per part: 0 not handed out, 1 handed out, 2 released
    void* mu_ = nullptr;
};
// Errors are Generate's (pattern.go:87-89); partSize must be positive.  The
// caller owns the returned object (delete it, or Close() and delete).
std::pair<GeneratedParts*, Error> GenerateParts(int64_t dataSize, int64_t workerID, const std::string& bucketName,
                                                const object::Object* obj, int64_t partSize);

// pattern.go:94-116
std::pair<std::vector<uint8_t>, Error> generateDataUnit(int64_t unitCount, int64_t workerID,
                                                        const std::string& bucketName,
                                                        const object::Object* obj);

// pattern.go:118-134
Error Valid(int64_t workerID, const std::string& expectedBucketName, const object::Object* obj,
            Reader& reader);

// Valid over a body that is already in memory: body[0, bodyLen) is what the
// reader would deliver.
Error ValidBytes(int64_t workerID, const std::string& expectedBucketName, const object::Object* obj,
                 const uint8_t* body, size_t bodyLen);

// Batch forms (no counterpart in the reference): the same results as calling
// Generate / Valid once per item, with one device launch per kind.
struct GenerateItem {
    int64_t dataSize = 0;
    int64_t workerID = 0;
    std::string bucketName;
    const object::Object* obj = nullptr;
    uint8_t* out = nullptr;            // dataSize bytes
};
struct ValidItem {
    int64_t workerID = 0;
    std::string expectedBucketName;
    const object::Object* obj = nullptr;
    const uint8_t* body = nullptr;     // what the reader would deliver
    size_t bodyLen = 0;
};
std::vector<Error> GenerateBatch(const std::vector<GenerateItem>& items);
std::vector<Error> ValidBatch(const std::vector<ValidItem>& items);

// Generate / Valid in two steps (no counterpart in the reference): Begin checks the arguments, hands the
// device work to the micro-batcher and returns at once; Poll says whether the call is over and, when it is,
// returns what Generate / Valid would have returned and frees the handle (true exactly once).  `out`
// (Generate) and `body` (Valid: the error text quotes the unit it found) must stay valid until then.  For a
// host language whose own tasks are cheaper to park than an OS thread is to put to sleep: two sleeps and two
// wake-ups per blocking call cost the host more than the device work takes.  With micro-batching off, or
// for a body over 256 KiB, Begin does the whole call and the first Poll returns true.
struct Pending;
Pending* GenerateBegin(int64_t dataSize, int64_t workerID, const std::string& bucketName, const object::Object* obj,
                       uint8_t* out);
Pending* ValidBegin(int64_t workerID, const std::string& expectedBucketName, const object::Object* obj,
                    const uint8_t* body, size_t bodyLen);
bool Poll(Pending* p, Error* result);

// pattern.go:136-180
Error validDataUnit(int64_t unitCount, int64_t workerID, const std::string& expectedBucketName,
                    const object::Object* obj, const uint8_t* data /* 256 bytes */);

// pattern.go:182-206 and encoding/hex.Dump
std::string dump(const std::string& data);
std::string hexDump(const uint8_t* data, size_t n);

// Engine control (no counterpart in the reference).
int  Init(int deviceID);            // oval_status; one engine on that GPU (OVAL_HIP_ENGINES=N or "all" in the environment: more)
// One engine per entry of `devices` (a GPU may be named more than once); calls are routed by worker ID,
// so a worker's objects always meet the same GPU.  The first Init / InitEngines call decides.
int  InitEngines(const int* devices, int n);
int  EngineCount();
void Shutdown();
void SetClock(int64_t unixMicro);   // >= 0 pins time.Now().UnixMicro(); < 0 = wall clock
void Seed(uint64_t seed);           // source of rand.Float64() in DecideSize
std::string LastEngineError();
// Micro-batching: while enabled, concurrent Generate / Valid calls (one per
// worker thread, internal/runner/runner.go:186-214) that arrive within
// `windowMicros` of each other share one device launch; with window 0 a batch is
// whatever queued up while the previous batch was on the device.  Results are
// identical either way.  A negative window keeps the current setting.
void SetMicroBatching(bool enable, int windowMicros = 0, size_t maxBytes = 64u << 20);
void MicroBatchingStats(uint64_t* batches, uint64_t* jobs);
// Batches in flight at once (1..8) and whether a lane naps between looks at its event
// (1) or polls (0) while its batch is on the device; negative / zero lanes keep
// the current value.  Applies from the next SetMicroBatching(true).
void SetBatchLanes(int lanes, int blockingSync);
// How the callers of a finished batch are woken (the lane wakes `roots` of them, each wakes `fanout`
// more; defaults 1 and 4) and how long a caller polls before it sleeps (default 0).  Values < 0 keep
// the setting.  For benchmarks.
void SetWakeTuning(int fanout, int roots, int spinNanos);
// How long a caller of a shared batch polls for its outcome before it sleeps (0 = never, the default; -1 = about as long as
// shared batches have lately taken) and how many callers may poll at once
// (-1 = a quarter of the cores that direct launches may use).
void SetCallerSpin(int nanos, int maxSpinners);
// Test hook: the next n small batches are launched so that their completion is never seen (as on a
// device that hangs); they end by the device time limit (oval_set_device_timeout_ms) and cost their
// staging slots, which QuarantinedSlots counts.
void TestLoseFlags(int n);
// Where the engines' staging slots stand (a snapshot taken without locks), for a tool whose calls are stuck.
void DebugDump(FILE* out);
int  QuarantinedSlots();
// How GenerateBatch / ValidBatch cut a large batch into sub-batches that travel through the device one
// behind the other: at most `chunkBytes` each (0 keeps the setting; default 16 MiB), `depth` in flight
// (0 keeps it; default 3).  For benchmarks.
void SetBatchShape(size_t chunkBytes, int depth);
// While micro-batching is on, a call that finds at most `callers` calls under way (itself included) still
// launches on its own: below that the hand-over to a lane costs more than it saves (default: the host's cores, at most 16; 0 = never).
void SetDirectCallers(int callers);
// Whether a small batch signals its end through a flag its kernel writes to pinned host memory (default)
// or through a stream event.  For benchmarks.
void SetFlagCompletion(bool on);
// Time the lane threads spent per phase since the last reset, summed over
// lanes: [0] ns waiting for callers still copying their bodies in, [1] ns
// enqueueing the batch, [2] ns waiting for the device, [3] ns publishing and
// waking, [4] ns re-taking the pipeline lock, [5] batches, [6] body bytes.
void LaneProfile(uint64_t out[8], bool reset);
// Valid(reader) consumes the body in chunks of at most this many bytes.
void SetStreamChunkBytes(size_t bytes);

}  // namespace pattern
}  // namespace oval
