// oval_engine.hpp — the device pipeline behind the host mirror of
// internal/pattern (oval_pattern.cpp): staging slots, the micro-batcher that
// folds concurrent single-object calls (one per worker goroutine,
// internal/runner/runner.go:186-214; internal/runner/worker.go:64-200) into
// shared launches, and the double-buffered chunk stream behind Valid(reader).
// Internal to the library.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/oval_hip.h"

namespace oval {
namespace engine {

// One object's worth of device work.
struct Job {
    bool is_fill = true;
    oval_obj_desc desc{};          // body_offset is assigned by the engine
    uint8_t* out = nullptr;        // fill: n_units * 256 bytes of host memory that receive the body
    const uint8_t* in = nullptr;   // validate: n_units * 256 bytes of host memory that hold the body
    oval_obj_result res{};         // validate: the device's verdict
    bool failed = false;           // the engine (not the data) failed; `err` says how
    std::string err;
};

// One engine per entry of `devices` (1..16; the same device may appear more than once).  The first call
// decides; later calls only check that the engines' devices answer.  oval_status.
int  init_engines(const int* devices, int n);
// One engine on `device`; with OVAL_HIP_ENGINES=N in the environment N engines on devices device,
// device + 1, ... modulo the visible GPUs ("all": one per visible GPU).  Idempotent.
int  init(int device);
int  engine_count();
void shutdown();
std::string last_error();          // text of the most recent engine failure

// Runs the jobs as one batch in a staging slot of their own and waits.
void run(Job* const* jobs, size_t n);
// One job: through the micro-batcher while it is enabled (and the job is small
// enough to share a slot), else as a batch of one.
void execute(Job& job);

// The same without a caller waiting inside the library.  submit() places the job in the micro-batcher's
// open slot (the validate body is copied then; `out` must stay valid until the job has been collected) and
// returns 1.  0: no room right now (every slot busy, the open one full): nothing was done, try again later —
// never by waiting inside the library, since the caller may itself hold the jobs that keep the slots busy.
// -1: the job cannot travel this way at all (micro-batching off, the job too large): call execute().  try_complete()
// returns false while the batch is under way; true once, with the outcome in `job`, after which the
// ticket is spent.  Every submitted job must be collected: its staging slot is held until then.
struct Ticket {
    void* slot = nullptr;
    uint32_t batch_seq = 0, idx = 0;
    size_t off = 0;
};
int submit(Job& job, Ticket& t);
bool try_complete(Ticket& t, Job& job);

void set_batching(bool enable, int window_us, size_t max_bytes);
// Lanes = batches in flight at once (1..8); blocking = lanes look at their batch's event in naps of about
// 15 us while it is on the device instead of polling it.  Takes effect when batching is next enabled.
void set_lanes(int lanes, int blocking);
// Shape of the tree through which the callers of a finished batch wake each other: the lane wakes
// `roots` callers, each caller wakes `fanout` more (defaults 1 and 4; values < 1 keep the setting).
void set_wake_tree(int fanout, int roots);
// How run() cuts a large batch: sub-batches of at most `chunk_bytes` (0 keeps the setting), `depth`
// of them in flight (1..6, 0 keeps it).  Defaults 16 MiB and 3.  For benchmarks.
void set_run_shape(size_t chunk_bytes, int depth);
// A call that finds at most this many calls under way (itself included) launches on its own instead of
// going through the micro-batcher (default: the host's cores, at most 16; 0 = always through the micro-batcher while it is on).
void set_direct_max(int callers);
// Whether a small batch on a polling lane signals its end through a flag its kernel writes to pinned
// host memory (default) or through a stream event.  For benchmarks.
void set_flag_completion(bool on);
// A caller of a shared batch polls its wake word this long before it sleeps on it: 0 (default) = sleeps at
// once, -1 = about as long as shared batches have lately taken, > 0 = that many ns.  At most `callers`
// callers poll at any moment (set_spin_budget; -1 = default, a quarter of the cores direct launches may
// use); the rest sleep at once.
void set_caller_spin(int ns);
void set_spin_budget(int callers);
// Test hook: the next `n` small batches that would signal their end through the flag in host memory are
// launched with a flag word nobody reads and are treated as never finishing, as a hung device would
// leave them: they end by the device time limit.
void test_lose_flags(int n);
// Staging slots set aside for good because their batch did not finish within the device time limit.
int  quarantined_slots();
void stats(uint64_t* batches, uint64_t* jobs);
// A lock-free snapshot of every engine's slots (tickets, readers, completion counters), for a tool that
// has found its calls stuck.
void debug_dump(FILE* out);
// Where the lane threads' time went since the last reset, summed over lanes:
// ns waiting for callers still copying in, ns enqueueing, ns waiting for the
// device, ns publishing + waking, ns re-taking the lock, batches, body bytes.
constexpr int kProfileWords = 8;
void lane_profile(uint64_t out[kProfileWords], bool reset);

// Double-buffered validation of one object's body as it arrives: the caller
// reads chunk k+1 into one pinned half while chunk k is on the device.
struct Stream;
Stream* stream_open(size_t chunk_bytes, uint32_t route, std::string* err);   // route: the worker ID (picks the engine)
uint8_t* stream_buffer(Stream* st, int half);              // pinned, chunk_bytes long
// Validates the first desc.n_units units of half `half` against `desc`
// (asynchronous); stream_wait returns that half's verdict.
int stream_submit(Stream* st, int half, const oval_obj_desc& desc);
int stream_wait(Stream* st, int half, oval_obj_result* res);
std::string stream_error(Stream* st);
void stream_close(Stream* st);

// A generated body that arrives in library-owned pinned memory slice by slice
// (what a multipart upload sends part by part, internal/s3client/s3client.go:164-236):
// parts_open starts the fill and returns at once; parts_wait blocks until bytes
// [begin, end) of the body have landed and returns the start of the buffer.
// The buffer comes from a pool of pinned allocations and goes back to it in
// parts_close.
struct Parts;
Parts* parts_open(const oval_obj_desc& desc, std::string* err);
const uint8_t* parts_wait(Parts* p, size_t begin, size_t end, std::string* err);   // NULL on engine failure
void parts_close(Parts* p);
// Idle pinned buffers the pool keeps / bytes they hold (for tests and tools).
void parts_pool_stats(size_t* buffers, size_t* bytes);

}  // namespace engine
}  // namespace oval
