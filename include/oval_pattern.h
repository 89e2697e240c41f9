/*
 * oval_pattern.h — C entry points of the host-side mirror of the reference's
 * internal/pattern and internal/object packages.
 *
 * The mirror itself is C++ (oval_amd/csrc/oval_pattern.hpp: namespaces
 * oval::object and oval::pattern with the reference's function names).  These
 * wrappers exist so that a language without a C++ FFI (ctypes in the tests, or
 * cgo) can call it; each one names the reference function it stands for.
 * Names are passed as pointer + length because Go strings may hold any bytes.
 *
 * Return value convention for the functions that return a Go `error` in the
 * reference: 0 = nil, 1 = an error whose text (exactly the reference's) was
 * written to `err` (NUL-terminated, truncated to err_cap), negative = the
 * engine itself failed (oval_status; text from oval_last_error()).
 */
#ifndef OVAL_PATTERN_H
#define OVAL_PATTERN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Creates the process-wide engine on `device_id` (idempotent).  With
 * OVAL_HIP_ENGINES=N in the environment: N engines on devices device_id,
 * device_id + 1, ... modulo the visible GPUs ("all": one per visible GPU). */
int oval_pattern_init(int device_id);
/* One engine per entry of devices[0..n) (1 <= n <= 16; a GPU may be named more
 * than once).  Every call is routed by its worker ID modulo the engine count:
 * the workers of a process carry consecutive IDs
 * (internal/runner/runner.go:110-113), so they spread evenly and all objects of
 * a worker meet the same GPU.  The first init call of a process decides. */
int oval_pattern_init_engines(const int* devices, int n);
int oval_pattern_engine_count(void);
void oval_pattern_shutdown(void);
/* oval_set_device_timeout_ms (oval_hip.h) under this header's naming: how long
 * a call waits for the device before it fails with a negative status ("device
 * did not finish the batch"); the staging slot of such a batch is never used
 * again, other calls go on.  Default 30000 ms; 0 = no limit. */
void oval_pattern_set_device_timeout(uint64_t ms);

/* time.Now().UnixMicro() as Generate sees it (pattern.go:103-104).  A value
 * >= 0 pins the clock (every unit of a call gets this value); < 0 returns to
 * the wall clock, sampled once per Generate call. */
void oval_pattern_set_clock(int64_t unix_micro);
/* Seeds the generator behind DecideSize's rand.Float64() (pattern.go:63). */
void oval_pattern_seed(uint64_t seed);

/* pattern.DecideSize — pattern.go:22-71 */
int oval_pattern_decide_size(int64_t min_size, int64_t max_size, int64_t* size_out,
                             char* err, size_t err_cap);
/* the same with the uniform sample y in [0,1) supplied by the caller */
int oval_pattern_decide_size_y(int64_t min_size, int64_t max_size, double y, int64_t* size_out,
                               char* err, size_t err_cap);

/* pattern.Generate — pattern.go:73-92.  out must hold data_size bytes. */
int oval_pattern_generate(int64_t data_size, int64_t worker_id,
                          const char* bucket, size_t bucket_len,
                          const char* key, size_t key_len, int64_t write_count,
                          uint8_t* out, char* err, size_t err_cap);

/* pattern.generateDataUnit — pattern.go:94-116.  out must hold 256 bytes. */
int oval_pattern_generate_data_unit(int64_t unit_count, int64_t worker_id,
                                    const char* bucket, size_t bucket_len,
                                    const char* key, size_t key_len, int64_t write_count,
                                    uint8_t* out, char* err, size_t err_cap);

/* pattern.Valid — pattern.go:118-134.  The io.Reader is the byte range
 * body[0, body_len); obj.Size is `size`. */
int oval_pattern_valid(int64_t worker_id, const char* bucket, size_t bucket_len,
                       const char* key, size_t key_len, int64_t size, int64_t write_count,
                       const uint8_t* body, size_t body_len, char* err, size_t err_cap);

/* pattern.Valid with a real io.Reader: `read` is called like Reader.Read
 * (returns the number of bytes stored in p, 0 for EOF or error) until
 * obj.Size / 256 units have been consumed or a bad unit was found.  The body
 * is validated chunk by chunk (oval_pattern_set_stream_chunk, default 32 MiB),
 * so its full size never has to be resident anywhere. */
typedef size_t (*oval_read_fn)(void* user, uint8_t* p, size_t n);
int oval_pattern_valid_stream(int64_t worker_id, const char* bucket, size_t bucket_len,
                              const char* key, size_t key_len, int64_t size, int64_t write_count,
                              oval_read_fn read, void* user, char* err, size_t err_cap);
void oval_pattern_set_stream_chunk(size_t bytes);

/* pattern.Generate for a body that is uploaded in parts.  The reference's
 * multipartUpload (internal/s3client/s3client.go:164-236) re-slices the []byte
 * that Generate returned (body[:partSize] at :185, body = body[partSize:] at
 * :204).  Here the body is generated ONCE into pinned host memory owned by the
 * library (no copy into caller memory) and part views of part_size bytes are
 * handed out of it; the body arrives there slice by slice, so part 0 can be on
 * the wire while later parts are still being copied from the device.
 * oval_parts_get waits until part `index` (bytes [index * part_size, min((index
 * + 1) * part_size, data_size))) has landed and returns a pointer into the
 * buffer, valid until that part is released.  When every part has been released
 * the buffer returns to the library's pool of pinned allocations;
 * oval_parts_close releases whatever is still held (the abort path,
 * s3client.go:190-203) and frees the handle, and must be called exactly once.
 * Return values as for oval_pattern_generate. */
typedef struct oval_parts oval_parts;
int oval_pattern_generate_parts(int64_t data_size, int64_t worker_id,
                                const char* bucket, size_t bucket_len,
                                const char* key, size_t key_len, int64_t write_count,
                                int64_t part_size, oval_parts** out, char* err, size_t err_cap);
int64_t oval_parts_count(const oval_parts* parts);     /* ceil(data_size / part_size) */
int oval_parts_get(oval_parts* parts, int64_t index, const uint8_t** data, size_t* len,
                   char* err, size_t err_cap);
int oval_parts_release(oval_parts* parts, int64_t index);
void oval_parts_close(oval_parts* parts);
/* Idle buffers of the pinned pool and the bytes they hold. */
void oval_parts_pool_stats(size_t* buffers, size_t* bytes);

/* pattern.validDataUnit — pattern.go:136-180.  data holds 256 bytes. */
int oval_pattern_valid_data_unit(int64_t unit_count, int64_t worker_id,
                                 const char* bucket, size_t bucket_len,
                                 const char* key, size_t key_len, int64_t write_count,
                                 const uint8_t* data, char* err, size_t err_cap);

/* Batch forms of Generate / Valid (no counterpart in the reference): item i
 * gets rcs[i] (0 nil, 1 error, <0 engine failure) and its error text at
 * errs + i * err_stride.  Returns the lowest (most severe) engine status. */
int oval_pattern_generate_batch(size_t n, const int64_t* data_sizes, const int64_t* worker_ids,
                                const char* const* buckets, const size_t* bucket_lens,
                                const char* const* keys, const size_t* key_lens,
                                const int64_t* write_counts, uint8_t* const* outs,
                                int* rcs, char* errs, size_t err_stride);
int oval_pattern_valid_batch(size_t n, const int64_t* worker_ids, const char* const* buckets,
                             const size_t* bucket_lens, const char* const* keys, const size_t* key_lens,
                             const int64_t* sizes, const int64_t* write_counts,
                             const uint8_t* const* bodies, const size_t* body_lens,
                             int* rcs, char* errs, size_t err_stride);

/* Generate / Valid in two steps (no counterpart in the reference), for a host
 * whose own tasks are cheaper to park than an OS thread is to put to sleep: a
 * goroutine that calls *_begin, yields to the Go scheduler between polls and
 * collects with oval_pattern_poll never blocks its thread in the library, and the
 * two sleeps and two wake-ups of a blocking call (which cost a 16-core host more
 * than the device work takes) are not paid.  *_begin checks the arguments, hands
 * the device work to the micro-batcher (oval_pattern_set_batching must be on) and
 * returns OVAL_OK with a handle.  oval_pattern_poll returns
 * OVAL_PATTERN_PENDING while the call is under way, else once the value the
 * blocking call would have returned (0 nil, 1 error text in err, < 0 engine
 * failure), and frees the handle.  `out` / `body` must stay valid until then
 * (Go: pinned with runtime.Pinner).  Every handle must be polled to its end: a
 * staging slot is held until its jobs have been collected (oval_pattern_shutdown
 * waits for such slots for the device time limit, at least a second, then leaves
 * them untouched; their handles can still be polled afterwards).  A call that cannot
 * share a launch (micro-batching off, body over 256 KiB) is carried out inside
 * *_begin; its first poll returns the result. */
#define OVAL_PATTERN_PENDING 2
typedef struct oval_pending oval_pending;
int oval_pattern_generate_begin(int64_t data_size, int64_t worker_id, const char* bucket, size_t bucket_len,
                                const char* key, size_t key_len, int64_t write_count, uint8_t* out,
                                oval_pending** pending);
int oval_pattern_valid_begin(int64_t worker_id, const char* bucket, size_t bucket_len,
                             const char* key, size_t key_len, int64_t size, int64_t write_count,
                             const uint8_t* body, size_t body_len, oval_pending** pending);
int oval_pattern_poll(oval_pending* pending, char* err, size_t err_cap);

/* Micro-batching of concurrent single-object calls (worker goroutines call
 * Generate / Valid one object at a time, internal/runner/worker.go:64-200):
 * while enabled, calls arriving within window_us share one device launch. */
void oval_pattern_set_batching(int enable, int window_us, size_t max_bytes);
/* Batches in flight at once (1..8, default 2) and whether a lane thread naps
 * between looks at its batch's event (1: about 15 us at a time, which leaves its
 * core to the callers) or polls (0, default) while the batch is on the device.
 * Values <= 0 / < 0 keep the current setting; applies from the next enable. */
void oval_pattern_set_lanes(int lanes, int blocking_sync);
/* Benchmarks only: the callers of a finished batch wake each other through a tree
 * (the lane wakes `roots` callers, each wakes `fanout` more; defaults 1 and 4), and
 * a caller may poll for `spin_ns` nanoseconds before it sleeps (default 0).
 * Negative values keep the current setting. */
void oval_pattern_set_wake_tuning(int fanout, int roots, int spin_ns);
/* While micro-batching is on, a call that finds at most `callers` calls under way
 * (itself included) still launches on its own: below that the hand-over to a lane
 * costs more than a shared launch saves (default: the host's cores, at most 16; 0 = every call goes through
 * the micro-batcher). */
void oval_pattern_set_direct_callers(int callers);
/* How long a caller of a shared launch polls for its outcome before it sleeps
 * (0, the default: never; -1: about as long as shared launches have lately
 * taken) and how many callers may poll at the same time (-1: a quarter of the
 * cores that calls launching on their own may use). */
void oval_pattern_set_caller_spin(int spin_ns, int max_spinners);
/* Test hook: the next n small launches never report their end, as on a device
 * that hangs; their calls fail after the device time limit and their staging
 * slots are set aside (counted by oval_pattern_quarantined_slots). */
void oval_pattern_test_lose_flags(int n);
int oval_pattern_quarantined_slots(void);
/* Launches and the jobs they carried since the library was loaded (a call that
 * launched on its own counts as a launch of one job). */
void oval_pattern_batching_stats(uint64_t* batches, uint64_t* jobs);

/* pattern.dump — pattern.go:182-206; and encoding/hex.Dump which feeds it.
 * Both return the full length of the text and write at most cap-1 bytes + NUL. */
size_t oval_pattern_dump(const char* text, size_t n, char* out, size_t cap);
size_t oval_pattern_hex_dump(const uint8_t* data, size_t n, char* out, size_t cap);

/* object.generateKey — object.go:48-50.  Returns the key length. */
size_t oval_object_generate_key(int64_t obj_id, char* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* OVAL_PATTERN_H */
