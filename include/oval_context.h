/*
 * oval_context.h — C entry points for the object bookkeeping and the saved
 * execution context that feed the pattern engine with batches (C++ mirror in
 * oval_amd/csrc/oval_context.hpp):
 *   internal/object/object.go:27-139     ObjectMeta
 *   internal/runner/runner.go:26-35      ExecutionContext (JSON: runner.go:85-104, 250-273)
 *   internal/runner/worker.go:16-29      Worker, BucketWithObject
 * and, with no counterpart in the reference, the descriptor table of every
 * object a context says exists (one device batch re-validates a whole key
 * space after a blackout, README.md:200-236).
 *
 * Return value convention as in oval_pattern.h: 0 = nil, 1 = an error whose
 * text was written to `err`, negative = oval_status.
 */
#ifndef OVAL_CONTEXT_H
#define OVAL_CONTEXT_H

#include <stddef.h>
#include <stdint.h>

#include "oval_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oval_execctx oval_execctx;   /* runner.ExecutionContext */
typedef struct oval_objmeta oval_objmeta;   /* object.ObjectMeta */

/* ---- ExecutionContext ---------------------------------------------------- */
/* Runner.init of a fresh run — runner.go:106-125.  start_worker_id < 0 draws
 * rand.Intn(0x10000) as the reference does. */
int oval_execctx_new(const char* endpoint, const char* const* bucket_names, size_t n_buckets,
                     int64_t num_obj, int64_t num_worker, int64_t min_size, int64_t max_size,
                     int64_t start_worker_id, int64_t runner_id, oval_execctx** out,
                     char* err, size_t err_cap);
/* loadSavedContext + the load branch of Runner.init — runner.go:85-104,111-129 */
int oval_execctx_load(const char* json, size_t len, oval_execctx** out, char* err, size_t err_cap);
int oval_execctx_load_file(const char* path, oval_execctx** out, char* err, size_t err_cap);
/* json.Marshal(execContext) — runner.go:257.  Returns the full length; writes
 * at most cap bytes (no terminator). */
size_t oval_execctx_marshal(const oval_execctx* ec, char* out, size_t cap);
/* Runner.SaveContext — runner.go:250-273 */
int oval_execctx_save_file(const oval_execctx* ec, const char* path, char* err, size_t err_cap);
void oval_execctx_destroy(oval_execctx* ec);

/* fields of the context; any out pointer may be NULL */
void oval_execctx_info(const oval_execctx* ec, int64_t* num_obj, int64_t* num_worker, int64_t* min_size,
                       int64_t* max_size, int64_t* start_worker_id, size_t* n_workers, size_t* n_bucket_names);
size_t oval_execctx_endpoint(const oval_execctx* ec, char* out, size_t cap);
/* Worker.id = (StartWorkerID + i) % 0x10000 — runner.go:112 */
int64_t oval_execctx_worker_id(const oval_execctx* ec, size_t worker);
size_t oval_execctx_worker_buckets(const oval_execctx* ec, size_t worker);
size_t oval_execctx_bucket_name(const oval_execctx* ec, size_t worker, size_t bucket, char* out, size_t cap);
/* borrowed: valid until the context is destroyed; NULL when out of range */
oval_objmeta* oval_execctx_objmeta(oval_execctx* ec, size_t worker, size_t bucket);
uint64_t oval_execctx_count_existing(const oval_execctx* ec);

/* ---- ObjectMeta ------------------------------------------------------------
 * Objects are named by their index in ObjectList (what the reference's object
 * IDs are, object.go:52-55); -1 stands for a nil *Object. */
oval_objmeta* oval_objmeta_new(int64_t num_obj, int64_t key_id_offset);   /* NewObjectMeta — object.go:57-70 */
void oval_objmeta_free(oval_objmeta* om);                                  /* only for oval_objmeta_new's */
int64_t oval_objmeta_get_random_object(oval_objmeta* om);                  /* object.go:72-75 */
int oval_objmeta_register_to_existing_list(oval_objmeta* om, const char* key, size_t key_len,
                                           char* err, size_t err_cap);     /* object.go:77-92 */
int64_t oval_objmeta_pop_existing_random_object(oval_objmeta* om);         /* object.go:94-110 */
int64_t oval_objmeta_get_existing_random_object(oval_objmeta* om);         /* object.go:112-120 */
/* object.go:122-130: *exists_out = 0/1 */
int oval_objmeta_exist(const oval_objmeta* om, const char* key, size_t key_len, int* exists_out,
                       char* err, size_t err_cap);
/* object.go:132-134; each buffer holds 13 bytes */
void oval_objmeta_get_head_and_tail_key(const oval_objmeta* om, char* head13, char* tail13);
int oval_objmeta_tidy_up(oval_objmeta* om, char* err, size_t err_cap);      /* object.go:136-141 */
size_t oval_objmeta_num_objects(const oval_objmeta* om);
size_t oval_objmeta_num_existing(const oval_objmeta* om);
int64_t oval_objmeta_existing_id(const oval_objmeta* om, size_t i);         /* ExistingObjectIDs[i] */
int64_t oval_objmeta_key_id_offset(const oval_objmeta* om);
size_t oval_objmeta_key_prefix(const oval_objmeta* om, char* out, size_t cap);
/* ObjectList[index]; key13 holds 13 bytes.  Returns 0, or OVAL_E_RANGE. */
int oval_objmeta_get_object(const oval_objmeta* om, int64_t index, char* key13, int64_t* size,
                            int64_t* write_count);
/* what the worker does to an *Object: obj.Size = size, obj.WriteCount++ (worker.go:82-84), obj.Clear() */
int oval_objmeta_set_object(oval_objmeta* om, int64_t index, int64_t size, int64_t write_count);
/* seeds the source of rand.Intn behind the random picks */
void oval_object_seed_random(uint64_t seed);

/* The state a run leaves behind, without running one (no counterpart in the
 * reference): every object exists with probability `fraction`, Size =
 * DecideSize(MinSize, MaxSize) — pattern.go:22-71 — and WriteCount uniform in
 * [1, max_write_count], from a generator seeded with `seed`. */
int oval_execctx_populate(oval_execctx* ec, uint64_t seed, double fraction, int64_t max_write_count,
                          char* err, size_t err_cap);

/* ---- descriptor table of the existing objects ----------------------------- */
typedef struct oval_obj_ref {
    uint32_t worker;   /* index into the context's workers */
    uint32_t bucket;   /* index into that worker's buckets */
    int64_t  obj_id;   /* index into that ObjectMeta's ObjectList */
} oval_obj_ref;

/* Descriptors of the objects of workers [worker_begin, worker_end) that exist,
 * bodies laid out back to back from offset 0, in worker / bucket /
 * ExistingObjectIDs order.  *n_out = how many there are, *body_bytes_out = the
 * bytes their bodies span; descs_out / refs_out (either may be NULL) receive
 * the first min(cap, *n_out) entries. */
int oval_execctx_build_descs(const oval_execctx* ec, uint64_t unix_micro, size_t worker_begin,
                             size_t worker_end, oval_obj_desc* descs_out, oval_obj_ref* refs_out,
                             uint64_t cap, uint64_t* n_out, uint64_t* body_bytes_out,
                             char* err, size_t err_cap);

/* The same objects as 12-byte records plus one segment per worker and bucket that
 * has any, for oval_layout_reset_compact (the descriptors are then written on
 * the device).  *n_segs_out / *n_obj_out / *body_bytes_out are always set;
 * segs_out and recs_out are written when both are given and large enough.
 * Fails when an object's key does not follow from its ID (a hand-edited saved
 * context): such a context needs oval_execctx_build_descs. */
int oval_execctx_build_compact(const oval_execctx* ec, uint64_t unix_micro, size_t worker_begin,
                               size_t worker_end, oval_obj_segment* segs_out, uint64_t seg_cap,
                               oval_obj_rec* recs_out, uint64_t rec_cap, uint64_t* n_segs_out,
                               uint64_t* n_obj_out, uint64_t* body_bytes_out, char* err, size_t err_cap);

#ifdef __cplusplus
}
#endif
#endif /* OVAL_CONTEXT_H */
