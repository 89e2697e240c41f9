/*
 * oval_hip.h — C ABI of the MI355X (gfx950) data-pattern engine.
 *
 * This is the boundary a cgo file in the reference's internal/pattern package
 * binds.  The reference has no FFI of its own (it is built CGO_ENABLED=0,
 * Makefile:23); the entry points below replace the bodies of
 *
 *   pattern.Generate          internal/pattern/pattern.go:73-92
 *   pattern.generateDataUnit  internal/pattern/pattern.go:94-116
 *   pattern.Valid             internal/pattern/pattern.go:118-134
 *   pattern.validDataUnit     internal/pattern/pattern.go:136-180
 *
 * for whole batches of objects at a time.  Plain pointers and sizes only; no
 * C++ or torch types.  Every function returns OVAL_OK (0) or a negative
 * oval_status; on failure a human-readable message is available from
 * oval_last_error() on the calling thread.  Nothing throws across this ABI.
 *
 * All structures are little-endian and fixed-width.  No pointer passed in is
 * retained after the call returns unless the function says so.
 */
#ifndef OVAL_HIP_H
#define OVAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants of the data unit (pattern.go:17-20, object.go:13-19) ---- */
#define OVAL_UNIT_SIZE        256u  /* dataUnitSize */
#define OVAL_MAX_BUCKET_LEN   16u   /* object.MaxBucketNameLength */
#define OVAL_MAX_KEY_LEN      12u   /* object.MaxKeyLength */
#define OVAL_HEADER_SIZE      48u   /* 16 + 12 + dataUnitHeaderSizeWithoutBucketAndKey */
#define OVAL_NO_BAD_UNIT      0xFFFFFFFFu

/* Bits of a per-unit field mask.  Bits 0..4 are the five checks of
 * validDataUnit in the order it reports them (pattern.go:140-170); bit 5 is
 * this engine's opt-in extension: some byte of [48,256) differs from its own
 * offset.  The reference never looks at bytes >= 40. */
#define OVAL_BAD_BUCKET   0x01u
#define OVAL_BAD_KEY      0x02u
#define OVAL_BAD_WRITECNT 0x04u
#define OVAL_BAD_OFFSET   0x08u
#define OVAL_BAD_WORKER   0x10u
#define OVAL_BAD_FILLER   0x20u
#define OVAL_REF_FIELDS   0x1Fu

typedef enum oval_status {
    OVAL_OK            =  0,
    OVAL_E_INVALID     = -1,  /* bad argument (null, misaligned, out of range) */
    OVAL_E_HIP         = -2,  /* a HIP runtime call failed                    */
    OVAL_E_NOMEM       = -3,  /* host or device allocation failed             */
    OVAL_E_LAYOUT      = -4,  /* descriptors overlap / unsorted / misaligned  */
    OVAL_E_FIELD       = -5,  /* bucket+key do not format to exactly 28 bytes */
    OVAL_E_RANGE       = -6,  /* body buffer smaller than the layout needs    */
    OVAL_E_NODEVICE    = -7,  /* no gfx950 device visible                     */
    OVAL_E_INTERNAL    = -8   /* a C++ exception was stopped at the boundary  */
} oval_status;

/*
 * One object of a batch: the expected state the reference keeps in
 * object.Object (object.go:21-25) plus where its body lives in the batch
 * buffer.  Bytes [0,48) are literally the header of the object's unit 0 as
 * generateDataUnit writes it (pattern.go:95-107), except that [32,36) holds
 * first_unit instead of a byte offset.
 */
typedef struct oval_obj_desc {
    uint8_t  bucket[16];   /* "%-16s" of the (<=16-byte) bucket name           */
    uint8_t  key[12];      /* "%-12s" of obj.Key                                */
    uint32_t write_count;  /* uint32(obj.WriteCount)                            */
    uint32_t first_unit;   /* unit index (unitCount) of this body's first unit;
                              0 for a whole object, >0 for a part of a stream   */
    uint32_t worker_id;    /* uint32(workerID)                                  */
    uint64_t unix_micro;   /* pinned time.Now().UnixMicro(); never validated    */
    uint64_t body_offset;  /* byte offset of the first unit in the batch buffer;
                              multiple of 256                                   */
    uint32_t n_units;      /* number of 256-byte units to fill / validate       */
    uint8_t  bucket_len;   /* byte length of the expected bucket name (<=16)    */
    uint8_t  key_len;      /* byte length of the expected key (<=12)            */
    uint16_t flags;        /* OVAL_DESC_* (set by oval_desc_init)               */
} oval_obj_desc;           /* 64 bytes */

/* The expected name has leading or trailing Unicode white space, so
 * strings.TrimSpace(field) can never equal it (pattern.go:140,147): every unit
 * must take the exact TrimSpace check instead of the 128-bit compare. */
#define OVAL_DESC_BUCKET_SLOW 0x0001u
#define OVAL_DESC_KEY_SLOW    0x0002u
/* The expected worker ID does not fit a uint32: validDataUnit compares it as a
 * Go int (pattern.go:166-167), so no unit can ever match. */
#define OVAL_DESC_WORKER_OOR  0x0004u
/* The names hold multi-byte runes, so "%-16s%-12s" pads them past 28 bytes and
 * the reference's Generate fails its length check (pattern.go:87-89).  Such an
 * object can be validated (the fields are compared as bytes) but not filled. */
#define OVAL_DESC_NO_FILL     0x0008u

/* Per-object outcome of a validate pass.  first_bad_unit is the unitCount at
 * which Valid would have returned its error (pattern.go:128-131): the lowest
 * unit with any of the five reference fields wrong, OVAL_NO_BAD_UNIT if none.
 * The first two words form one little-endian u64 (unit in the high half). */
typedef struct oval_obj_result {
    uint32_t first_bad_fields;      /* OVAL_BAD_* (bits 0..4) of that unit; 0xFFFFFFFF if none */
    uint32_t first_bad_unit;        /* reference semantics                         */
    uint32_t bad_units;             /* units with a reference-visible mismatch     */
    uint32_t strict_first_bad_unit; /* lowest unit with ANY mismatch outside the
                                       time bytes [40,48), filler included         */
} oval_obj_result;                  /* 16 bytes */

/* Batch-wide counters: the vector that is summed across GPUs. */
enum {
    OVAL_CNT_OBJECTS_CHECKED = 0,
    OVAL_CNT_OBJECTS_BAD     = 1,
    OVAL_CNT_UNITS_CHECKED   = 2,
    OVAL_CNT_UNITS_BAD       = 3,
    OVAL_CNT_BAD_BUCKET      = 4,
    OVAL_CNT_BAD_KEY         = 5,
    OVAL_CNT_BAD_WRITECNT    = 6,
    OVAL_CNT_BAD_OFFSET      = 7,
    OVAL_CNT_BAD_WORKER      = 8,
    OVAL_CNT_STRICT_UNITS_BAD = 9,
    OVAL_CNT_RESERVED0       = 10,
    OVAL_CNT_RESERVED1       = 11,
    OVAL_NUM_COUNTERS        = 12
};

typedef struct oval_ctx    oval_ctx;     /* one device + one stream          */
typedef struct oval_layout oval_layout;  /* device-resident descriptor table */

/* ---- process-wide ------------------------------------------------------ */
const char* oval_last_error(void);            /* thread-local, never NULL     */
const char* oval_version(void);
int oval_device_count(int* n_out);
/* How long a host thread inside this library waits for the device before it
 * gives up: the host-memory entry points below and everything behind
 * oval_pattern.h fail with OVAL_E_HIP ("the device did not finish within the
 * time limit") instead of waiting for ever on a kernel or copy that never
 * ends.  Default 30000; 0 = no limit.  oval_ctx_sync and oval_timer_elapsed_ms
 * are the caller's own waits and stay unbounded.  The reference has no
 * counterpart (its pattern code cannot hang). */
int oval_set_device_timeout_ms(uint64_t ms);

/* ---- context ------------------------------------------------------------
 * A context binds one device and one stream.  `stream` is a hipStream_t cast
 * to void*; NULL makes the context create (and own) a non-blocking stream.
 * A context is used by one host thread at a time; several contexts may run
 * concurrently (one per goroutine-pinned OS thread, or one per GPU). */
int oval_ctx_create(int device_id, void* stream, oval_ctx** out);
int oval_ctx_destroy(oval_ctx* ctx);
int oval_ctx_sync(oval_ctx* ctx);
int oval_ctx_stream(oval_ctx* ctx, void** stream_out);

/* ---- memory the host side may need --------------------------------------
 * Pinned host buffers are what a cgo caller wraps with unsafe.Slice: Go memory
 * may not be handed to an asynchronous copy. */
int oval_dev_alloc(oval_ctx* ctx, size_t bytes, void** dptr_out);
int oval_dev_free(oval_ctx* ctx, void* dptr);
int oval_pinned_alloc(oval_ctx* ctx, size_t bytes, void** hptr_out);
int oval_pinned_free(oval_ctx* ctx, void* hptr);
int oval_copy_h2d(oval_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes); /* async on the ctx stream */
int oval_copy_d2h(oval_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes); /* async on the ctx stream */

/* ---- descriptors --------------------------------------------------------
 * Host-side helper that applies the reference's formatting rules: bucket is
 * cut to its first 16 bytes (pattern.go:74-76,119-121) and both names are
 * left-justified and space padded like fmt's "%-16s%-12s" (pattern.go:95,98).
 * Returns OVAL_E_FIELD for a key longer than 12 bytes ("%-12s" never
 * truncates, so no 256-byte unit exists for it) and sets OVAL_DESC_NO_FILL for
 * names whose rune count differs from their byte count. write_count and worker_id are truncated to 32 bits as
 * the reference's uint32() conversions do (pattern.go:101,105). */
int oval_desc_init(oval_obj_desc* d,
                   const char* bucket, size_t bucket_len,
                   const char* key, size_t key_len,
                   int64_t write_count, int64_t worker_id, uint64_t unix_micro,
                   uint64_t body_offset, uint32_t first_unit, uint32_t n_units);

/* ---- layout -------------------------------------------------------------
 * Uploads `n_obj` descriptors and builds the device-side chunk index.
 * Requirements (checked; OVAL_E_LAYOUT otherwise): body_offset is a multiple
 * of 256 and bodies are in ascending, non-overlapping order (gaps and
 * zero-unit objects are allowed); at most 2^30 objects and just under 1 TiB of
 * bodies per batch.  The layout can be used for any number of
 * fill / validate passes and owns all device memory it needs, so those passes
 * allocate nothing.  The descriptor array is copied before the call returns;
 * the upload and the index build run asynchronously on the context's stream,
 * and a pass issued through another context of the same device waits for them. */
int oval_layout_create(oval_ctx* ctx, const oval_obj_desc* descs_host, uint64_t n_obj,
                       oval_layout** out);
/* Replaces the contents of an existing layout with a new descriptor set,
 * reusing its device memory when it is large enough: then nothing is allocated
 * and the device is not waited for (the new tables are written in stream order
 * behind the passes already issued on this context).  When it has to grow, it
 * waits for those passes first.  No pass that uses the layout may be in flight
 * on another context.  The OVAL_DESC_BUCKET_SLOW / KEY_SLOW bits of the copy
 * are derived from the names, whatever the caller's flags say. */
int oval_layout_reset(oval_ctx* ctx, oval_layout* layout, const oval_obj_desc* descs_host,
                      uint64_t n_obj);
/* oval_layout_reset for a job whose every batch brings new descriptors (the
 * worker loop of internal/runner/worker.go:38-121 produces new sizes and write
 * counts with every PUT): the upload and the index build run on a second stream
 * of the context, behind the last pass that used THIS layout and nothing else,
 * so with two layouts taken in turn the tables of batch k+1 travel (64 bytes
 * per object over PCIe: for 4k-16k objects as long as a pass over the bodies
 * takes) while the passes of batch k run.  Passes issued afterwards through
 * this context wait for the tables by themselves.  Passes over the layout
 * issued through OTHER contexts are the caller's to order against the rebuild.
 * Otherwise as oval_layout_reset. */
int oval_layout_reset_ahead(oval_ctx* ctx, oval_layout* layout, const oval_obj_desc* descs_host,
                            uint64_t n_obj);
/* ---- layouts from compact records ---------------------------------------
 * What the reference keeps per object is a key that follows from the object ID
 * (object.go:48-50: "ov" + 10 hex digits), a size and a write count
 * (object.go:21-25); bucket, worker ID and key range belong to the worker
 * (runner.go:117-125).  A batch whose bodies lie back to back can therefore be
 * described by 12 bytes per object plus one segment per run of objects that
 * share the rest, and the 64-byte descriptors are written on the device: a fifth
 * of the bytes over PCIe and no table to build, check and stage on the host.
 * Record i belongs to the segment with first_obj <= i < first_obj + n_obj; the
 * segments cover records 0 .. n_obj-1 in order.  Object i starts where object
 * i-1 ends (object 0 at offset 0) and is described from its unit 0. */
typedef struct oval_obj_rec {
    uint32_t obj_id;       /* key = "ov" + %010x of (segment.key_id_base + obj_id)        */
    uint32_t write_count;  /* uint32(obj.WriteCount), pattern.go:101                     */
    uint32_t n_units;      /* obj.Size / 256                                             */
} oval_obj_rec;            /* 12 bytes */

typedef struct oval_obj_segment {
    uint8_t  bucket[16];   /* space padded, as oval_obj_desc.bucket                      */
    uint64_t key_id_base;  /* (runnerID << 32) + (worker index << 24), runner.go:119-124 */
    uint64_t unix_micro;
    uint32_t worker_id;
    uint32_t first_obj;    /* index of the segment's first record                        */
    uint32_t n_obj;
    uint8_t  bucket_len;
    uint8_t  reserved;
    uint16_t flags;        /* OVAL_DESC_* bits that hold for every object of the segment */
} oval_obj_segment;        /* 48 bytes */

/* Formats the bucket like oval_desc_init and derives the flags (a bucket with
 * white space at an edge, a worker ID outside uint32, a bucket whose rune count
 * differs from its byte count). */
int oval_segment_init(oval_obj_segment* seg, const char* bucket, size_t bucket_len, int64_t worker_id,
                      uint64_t unix_micro, uint64_t key_id_base, uint32_t first_obj, uint32_t n_obj);
/* oval_layout_reset from records and segments in host memory (copied before the
 * call returns); with OVAL_RESET_AHEAD in `flags` as oval_layout_reset_ahead
 * orders it.  OVAL_E_LAYOUT if the segments do not cover the records in
 * order, if a key would need more than 10 hex digits, or if the bodies exceed
 * the 1 TiB batch limit. */
#define OVAL_RESET_AHEAD 0x1u
int oval_layout_reset_compact(oval_ctx* ctx, oval_layout* layout, const oval_obj_segment* segs_host,
                              uint64_t n_segs, const oval_obj_rec* recs_host, uint64_t n_obj, uint32_t flags);
/* Copies descriptors [first, first + n) of a layout's device table back to the
 * host and waits (tests and debugging: the table of a compact layout exists
 * only on the device). */
int oval_layout_read_descs(oval_ctx* ctx, const oval_layout* layout, uint64_t first, uint64_t n,
                           oval_obj_desc* out_host);
int oval_layout_destroy(oval_ctx* ctx, oval_layout* layout);
int oval_layout_info(const oval_layout* layout, uint64_t* n_obj, uint64_t* total_units,
                     uint64_t* span_bytes);
/* How many launches of the fill and of the validate kernel one pass over the
 * layout takes (a pass is cut where a launch would exceed the HIP grid limit
 * or depend on more lookup-table bytes than stay cached); for callers that
 * relate a profiler's per-launch durations to a pass. */
int oval_layout_launches(const oval_layout* layout, uint32_t* fill_launches, uint32_t* validate_launches);

/* ---- the two kernels ----------------------------------------------------
 * oval_fill: writes every unit of every object of the layout into body_dev
 * (device memory, 256-byte aligned, >= span_bytes long).  Replaces the loop
 * of Generate over generateDataUnit (pattern.go:79-85,94-116).
 *
 * oval_validate: reads every unit once, regenerates the expected unit in
 * registers and writes one oval_obj_result per object (device memory,
 * results_dev[n_obj]) plus, if counters_dev is not NULL, ADDS the batch totals
 * into counters_dev[OVAL_NUM_COUNTERS] (device memory, caller zeroes it).
 * Replaces the loop of Valid over validDataUnit (pattern.go:123-132,136-180).
 * Both are asynchronous on the context's stream.  The body may be any memory
 * the GPU can address: given a buffer from oval_pinned_alloc the kernels' own
 * stores and loads cross the link at its rate, with no staging copy. */
int oval_fill(oval_ctx* ctx, const oval_layout* layout, void* body_dev, uint64_t body_bytes);
int oval_validate(oval_ctx* ctx, const oval_layout* layout, const void* body_dev,
                  uint64_t body_bytes, oval_obj_result* results_dev, uint64_t* counters_dev);

/* oval_validate for callers that cannot allocate (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart
 * in the reference.  Same results and counters as oval_layout_create +
 * oval_validate over the same table, but nothing is allocated or freed: the
 * descriptor table is the caller's, already in device memory (descs_dev[n_obj],
 * 64-byte aligned, bodies ascending and non-overlapping as oval_layout_create
 * requires; not checked here since the host never sees it: a table that breaks
 * the rule gives meaningless results, but nothing outside body_dev[0,
 * span_bytes), results_dev[0, n_obj) and counters_dev is touched), and the
 * chunk table and tallies live in a static device array of 4 MiB + 32 KiB that
 * each GPU the library runs on holds until the process ends.  span_bytes = end
 * of the last object's body, at most OVAL_NOALLOC_MAX_SPAN (OVAL_E_RANGE
 * beyond).  Asynchronous on the context's stream.  Every call on a GPU shares
 * that GPU's one array: calls that overlap on one GPU (two contexts, two
 * streams) race; calls in stream order do not. */
#define OVAL_NOALLOC_MAX_SPAN (4ull << 30)
int oval_validate_noalloc(oval_ctx* ctx, const oval_obj_desc* descs_dev, uint64_t n_obj,
                          uint64_t span_bytes, const void* body_dev, uint64_t body_bytes,
                          oval_obj_result* results_dev, uint64_t* counters_dev);

/* Copies the 256 bytes at body_dev + byte_offset to out256 (host) and waits:
 * the unit the host needs to format the reference's error text
 * (pattern.go:141-175). */
int oval_read_unit(oval_ctx* ctx, const void* body_dev, uint64_t byte_offset, uint8_t* out256);

/* ---- one-shot helpers for bodies that start and end in host memory -------
 * body_host may be pinned (fast path) or pageable.  They create a temporary
 * layout, stage through device memory in chunks with copies overlapped on a
 * second stream, and wait for completion.  results_host[n_obj] and
 * counters_host[OVAL_NUM_COUNTERS] (may be NULL) are host memory. */
int oval_fill_batch_host(oval_ctx* ctx, const oval_obj_desc* descs_host, uint64_t n_obj,
                         void* body_host, uint64_t body_bytes);
int oval_validate_batch_host(oval_ctx* ctx, const oval_obj_desc* descs_host, uint64_t n_obj,
                             const void* body_host, uint64_t body_bytes,
                             oval_obj_result* results_host, uint64_t* counters_host);

/* ---- the one collective -----------------------------------------------------
 * Objects shard over GPUs by worker key range (internal/runner/runner.go:117-124:
 * key ID = (runnerID << 32) + (worker << 24) + j), one process or thread and one
 * context per GPU; the only thing that crosses GPUs is the sum of the counter
 * vectors that oval_validate adds to.  The reference has no counterpart (its
 * processes report to a leader over HTTP, internal/multiprocess).  RCCL is
 * opened when the first of these functions is called, not when the library
 * loads. */
#define OVAL_COMM_ID_BYTES 128
typedef struct oval_comm oval_comm;
/* One rank creates the ID and hands the 128 bytes to the others by any means. */
int oval_comm_unique_id(uint8_t* id_out);
/* Collective over all ranks: every rank calls it with the same ID, with its
 * context's device current. */
int oval_comm_create(oval_ctx* ctx, int n_ranks, int rank, const uint8_t* id, oval_comm** out);
int oval_comm_destroy(oval_comm* comm);
/* Size of the communicator and this rank's place in it, as RCCL reports them
 * (ncclCommCount, ncclCommUserRank).  Either output may be NULL. */
int oval_comm_info(const oval_comm* comm, int* n_ranks_out, int* rank_out);
/* counters_dev[0..n) = sum over all ranks, in place, asynchronously on the
 * context's stream (ncclAllReduce, ncclUint64, ncclSum). */
int oval_allreduce_counters(oval_ctx* ctx, oval_comm* comm, uint64_t* counters_dev, size_t n);

/* Optional second form of the exchange: the first-bad-unit table of the WHOLE job
 * on every rank (4 bytes per object: 40 MB at 10 M objects).  Each rank fills a
 * table of the job's object count with 0xFF bytes (OVAL_NO_BAD_UNIT), writes its
 * own objects' first_bad_unit at its own place with oval_gather_first_bad
 * (table_dev points at that place: table_dev[i] = results_dev[i].first_bad_unit),
 * and the minimum over ranks (ncclAllReduce, ncclUint32, ncclMin) leaves every
 * rank with the table of all objects.  Both asynchronous on the context's stream. */
int oval_gather_first_bad(oval_ctx* ctx, const oval_obj_result* results_dev, uint64_t n_obj, uint32_t* table_dev);
int oval_allreduce_first_bad(oval_ctx* ctx, oval_comm* comm, uint32_t* table_dev, size_t n);

/* ---- kernel launch tuning (benchmarks only) ------------------------------
 * Workgroup sizes of the fill and the validate kernel: 128, 256, 512 or 1024
 * threads; 0 restores the default (256, the fastest measured for both). */
int oval_ctx_set_tuning(oval_ctx* ctx, int fill_threads, int validate_threads);
/* Layouts whose objects average fewer than `fill_bytes` / `validate_bytes`
 * (span of the batch / number of objects) are filled / validated by the kernels
 * for objects of a few units, which resolve all objects of a 4 KiB chunk with one
 * vector load instead of walking them one dependent scalar load at a time (the
 * reference's smallest object is a single 256-byte unit, pattern.go:23-34).  The
 * defaults are where the two kinds of kernel measured alike on MI355X: one-unit
 * objects fill 2.8 x and validate 1.9 x faster with them, 4 KiB objects 10 %
 * slower.  Process-wide, read when a layout is created or reset; 0 = never,
 * UINT64_MAX = every layout.  Results are identical either way. */
#define OVAL_SMALL_FILL_BYTES_DEFAULT 3072u
#define OVAL_SMALL_VALIDATE_BYTES_DEFAULT 1280u
int oval_set_small_object_bytes(uint64_t fill_bytes, uint64_t validate_bytes);
/* Layouts whose objects average at least `validate_bytes` (multipart-sized
 * bodies, s3client.go:164-236) are validated by the kernel that reads with one
 * load per thread from workgroups of 1024 threads and looks its object up once
 * per workgroup.  0 = every layout, UINT64_MAX = never.  Results are identical
 * either way. */
#define OVAL_BIG_VALIDATE_BYTES_DEFAULT (1ull << 20)
int oval_set_big_object_bytes(uint64_t validate_bytes);

/* ---- timing --------------------------------------------------------------
 * HIP events recorded on the context's stream, for callers (bench.py) that
 * launch through this ABI and therefore cannot use another runtime's events. */
typedef struct oval_timer oval_timer;
int oval_timer_create(oval_ctx* ctx, oval_timer** out);
int oval_timer_destroy(oval_timer* t);
int oval_timer_start(oval_ctx* ctx, oval_timer* t);
int oval_timer_stop(oval_ctx* ctx, oval_timer* t);
int oval_timer_elapsed_ms(oval_timer* t, float* ms_out);  /* waits for stop */

#ifdef __cplusplus
}
#endif
#endif /* OVAL_HIP_H */
