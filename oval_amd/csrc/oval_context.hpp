// oval_context.hpp — host-side mirror of the object bookkeeping and of the saved
// execution context that feed the pattern engine with batches:
//   internal/object/object.go:27-139      ObjectMeta
//   internal/runner/runner.go:26-35       ExecutionContext
//   internal/runner/worker.go:16-29       Worker, BucketWithObject (serialised fields)
//   internal/runner/runner.go:85-135      loadSavedContext + Runner.init
//   internal/runner/runner.go:250-273     SaveContext
// plus what the reference has no counterpart for: turning the objects a context
// says exist into one descriptor table, so that re-validating a whole key space
// after a "blackout" (README.md:200-236) is one device batch.
//
// The JSON text is what Go's encoding/json produces and accepts for these
// structs (field names from the struct tags, compact output, HTML-safe string
// escaping; on input: any white space, any field order, case-insensitive field
// names, unknown fields ignored, null leaves a scalar as it is).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/oval_hip.h"
#include "oval_pattern.hpp"

namespace oval {

namespace object {

// object.go:27-33.  existingObjectIDMap (a Go map used as a set) is a byte per
// object here: IDs in it are indices into ObjectList (object.go:52-55 parses
// the 6 hex digits behind the key prefix), so membership is one load, and 10M
// objects cost 10 MB instead of a hash table.
class ObjectMeta {
public:
    std::vector<Object> ObjectList;
    std::vector<int64_t> ExistingObjectIDs;
    int64_t KeyIDOffset = 0;
    std::string KeyPrefix;

    ObjectMeta() = default;
    ObjectMeta(int64_t numObj, int64_t keyIDOffset);             // NewObjectMeta, object.go:57-70

    Object* GetRandomObject();                                   // object.go:72-75
    Error RegisterToExistingList(const std::string& key);        // object.go:77-92
    Object* PopExistingRandomObject();                           // object.go:94-110 (nullptr = nil)
    Object* GetExistingRandomObject();                           // object.go:112-120
    std::pair<bool, Error> Exist(const std::string& key) const;  // object.go:122-130
    std::pair<std::string, std::string> GetHeadAndTailKey() const;   // object.go:132-134
    Error TidyUp();                                              // object.go:136-141

private:
    std::vector<uint8_t> existing_;                              // existingObjectIDMap
};

// object.go:52-55.  The reference exits the process on a malformed key; here
// the error comes back to the caller.
std::pair<int64_t, Error> getObjIDFromKey(const std::string& key);

// Source of rand.Intn for the random picks above (process-wide, like Go's).
void SeedRandom(uint64_t seed);

}  // namespace object

namespace runner {

constexpr int maxWorkerID = 0x10000;                             // runner.go:22-24

struct BucketWithObject {                                        // worker.go:26-29
    std::string BucketName;
    object::ObjectMeta ObjectMeta;
};

struct Worker {                                                  // worker.go:16-24
    int64_t id = 0, minSize = 0, maxSize = 0;                    // not serialised; set by Init
    std::vector<BucketWithObject> BucketsWithObject;
};

struct ExecutionContext {                                        // runner.go:26-35
    std::string Endpoint;
    std::vector<std::string> BucketNames;
    int64_t NumObj = 0, NumWorker = 0, MinSize = 0, MaxSize = 0, StartWorkerID = 0;
    std::vector<Worker> Workers;
};

// Runner.init for a fresh run (runner.go:106-125): NumWorker workers, each with
// one ObjectMeta of NumObj/NumWorker objects per bucket and key IDs
// (runnerID << 32) + (worker << 24) + j.  startWorkerID < 0 draws it at random.
std::pair<ExecutionContext, Error> NewExecutionContext(const std::string& endpoint,
                                                       const std::vector<std::string>& bucketNames,
                                                       int64_t numObj, int64_t numWorker, int64_t minSize,
                                                       int64_t maxSize, int64_t startWorkerID, int64_t runnerID);

// loadSavedContext (runner.go:85-104) followed by the load branch of
// Runner.init (runner.go:111-129): worker ids and sizes restored, TidyUp on
// every ObjectMeta.
std::pair<ExecutionContext, Error> LoadSavedContext(const char* json, size_t len);
std::pair<ExecutionContext, Error> LoadSavedContextFile(const std::string& path);

// json.Marshal(execContext) (runner.go:257)
std::string MarshalContext(const ExecutionContext& ec);
Error SaveContext(const ExecutionContext& ec, const std::string& path);   // runner.go:250-273

// One entry per object that the context says exists, in worker, bucket,
// ExistingObjectIDs order.
struct ObjectRef {
    uint32_t worker = 0, bucket = 0;
    int64_t objID = 0;
};
struct DescriptorTable {
    std::vector<oval_obj_desc> descs;     // bodies back to back from offset 0
    std::vector<ObjectRef> refs;          // refs[i] names the object of descs[i]
    uint64_t bodyBytes = 0;
};
// workers [workerBegin, workerEnd) only (a rank's share, runner.go:117-124).
std::pair<DescriptorTable, Error> BuildExistingDescriptors(const ExecutionContext& ec, uint64_t unixMicro,
                                                           size_t workerBegin, size_t workerEnd);
// The same into the caller's arrays (either may be NULL; entries beyond `cap` are not written).  With
// both NULL only *nOut and *bodyBytesOut are computed: how large a table the caller has to provide.
Error BuildExistingDescriptorsInto(const ExecutionContext& ec, uint64_t unixMicro, size_t workerBegin, size_t workerEnd,
                                   oval_obj_desc* descs, ObjectRef* refs, size_t cap, size_t* nOut, uint64_t* bodyBytesOut);
// The same objects in compact form (12-byte records plus one segment per worker
// and bucket that has any: oval_layout_reset_compact turns them into descriptors
// on the device).  Fails if an object's key is not generateKey(KeyIDOffset + its
// index), which a hand-edited saved context can hold: such a context needs the
// descriptor form.  With recs and segs NULL only the counts are computed.
Error BuildExistingCompactInto(const ExecutionContext& ec, uint64_t unixMicro, size_t workerBegin, size_t workerEnd,
                               oval_obj_segment* segs, size_t segCap, oval_obj_rec* recs, size_t recCap,
                               size_t* nSegsOut, size_t* nObjOut, uint64_t* bodyBytesOut);
uint64_t CountExisting(const ExecutionContext& ec);

// Puts a context into the state a run leaves behind, without running one: every
// object exists with probability `fraction`, with Size = DecideSize(MinSize,
// MaxSize) (pattern.go:22-71) and WriteCount uniform in [1, maxWriteCount], from
// a generator seeded with `seed`.  The source of synthetic batches shaped like
// the reference's (worker.go:77-84).
Error PopulateExisting(ExecutionContext& ec, uint64_t seed, double fraction, int64_t maxWriteCount);

}  // namespace runner
}  // namespace oval
