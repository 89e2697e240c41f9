/*
 * oval_oracle.h — CPU restatement of the reference's data-pattern engine.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared
 * against; nothing under oval_amd/ may include, link or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * It restates, in plain C, internal/pattern/pattern.go and the constants of
 * internal/object/object.go of peng225/oval (paths below are relative to the
 * reference tree).  The Go toolchain is not available where this was written,
 * so the restatement is pinned by the reference's own known-answer data:
 * README.md:97-122 (one complete unit, one complete error message with dump)
 * and internal/pattern/pattern_test.go:30-255.  DecideSize's exact return
 * value for a given uniform sample and hex.Dump's partial-line layout have no
 * vector in the reference: parity unpinned for those two.
 */
#ifndef OVAL_ORACLE_H
#define OVAL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_UNIT_SIZE 256           /* pattern.go:18 dataUnitSize */
#define ORACLE_HDR_NUM   20            /* pattern.go:19 */
#define ORACLE_MAX_BUCKET 16           /* object.go:14 */
#define ORACLE_MAX_KEY    12           /* object.go:15 */

/* Same memory layout as the engine's descriptor/result so that one numpy
 * array can be handed to both sides; defined again here on purpose so the
 * oracle builds without the engine's headers. */
typedef struct oracle_desc {
    uint8_t  bucket[16];
    uint8_t  key[12];
    uint32_t write_count;
    uint32_t first_unit;
    uint32_t worker_id;
    uint64_t unix_micro;
    uint64_t body_offset;
    uint32_t n_units;
    uint8_t  bucket_len;
    uint8_t  key_len;
    uint16_t flags;
} oracle_desc;

typedef struct oracle_result {
    uint32_t first_bad_fields;
    uint32_t first_bad_unit;
    uint32_t bad_units;
    uint32_t strict_first_bad_unit;
} oracle_result;

#define ORACLE_NUM_COUNTERS 12

/* unicode/utf8 and strings helpers (Go standard library behaviour) */
size_t oracle_rune_count(const uint8_t* s, size_t n);
void   oracle_trim_space(const uint8_t* s, size_t n, size_t* start, size_t* stop);
/* fmt "%-<width>s": returns bytes written to out (cap >= n + width) */
size_t oracle_pad_field(const uint8_t* s, size_t n, int width, uint8_t* out);

/* object.go:48-50 generateKey: "ov" + %010x; out cap >= 24; returns length */
size_t oracle_generate_key(int64_t obj_id, char* out);

/* pattern.go:94-116; out cap >= blen + klen + 28 + 228; returns bytes written */
size_t oracle_generate_unit(int64_t unit_count, int64_t worker_id,
                            const uint8_t* bucket, size_t blen,
                            const uint8_t* key, size_t klen,
                            int64_t write_count, uint64_t unix_micro, uint8_t* out);

/* pattern.go:73-92.  Returns 0 and fills out[0..data_size) on success; returns
 * 1 and an error text otherwise (out untouched). */
int oracle_generate(int64_t data_size, int64_t worker_id,
                    const uint8_t* bucket, size_t blen,
                    const uint8_t* key, size_t klen,
                    int64_t write_count, uint64_t unix_micro,
                    uint8_t* out, char* err, size_t err_cap);

/* pattern.go:136-180.  Returns the mask of wrong fields (bit0 bucket, bit1 key,
 * bit2 write count, bit3 offset, bit4 worker; 0 = nil error) and, when err is
 * not NULL and the mask is non-zero, the exact error text. */
int oracle_valid_unit(int64_t unit_count, int64_t worker_id,
                      const uint8_t* exp_bucket, size_t blen,
                      const uint8_t* key, size_t klen,
                      int64_t write_count, const uint8_t* data,
                      char* err, size_t err_cap);

/* pattern.go:118-134 over an in-memory body of body_len bytes (the reader).
 * Returns 0 = nil, 1 = a unit failed (bad_unit set), 2 = short read (bad_unit
 * = index of the unit that could not be read). */
int oracle_valid(int64_t worker_id, const uint8_t* exp_bucket, size_t blen,
                 const uint8_t* key, size_t klen, int64_t size, int64_t write_count,
                 const uint8_t* body, size_t body_len,
                 int64_t* bad_unit, char* err, size_t err_cap);

/* encoding/hex.Dump and pattern.go:182-206 dump; return bytes written
 * (excluding the NUL that is always appended when cap allows). */
size_t oracle_hex_dump(const uint8_t* data, size_t n, char* out, size_t cap);
size_t oracle_dump(const char* text, size_t n, char* out, size_t cap);

/* pattern.go:22-71 with the uniform sample y = rand.Float64() injected. */
int oracle_decide_size(int64_t min_size, int64_t max_size, double y,
                       int64_t* size_out, char* err, size_t err_cap);

/* Batch forms: loops of the scalar functions above over descriptor arrays. */
void oracle_fill_batch(const oracle_desc* d, uint64_t n_obj, uint8_t* body);
void oracle_validate_batch(const oracle_desc* d, uint64_t n_obj,
                           const uint8_t* body, uint64_t body_bytes,
                           oracle_result* res, uint64_t* counters);

/* CPU-baseline timing variant: Generate then Valid of every object, paying per
 * unit what pattern.go:94-116 and :118-180 pay per unit (printf formatting,
 * small allocations, a clock read, append copies).  Returns units processed;
 * *bad_out = units that failed to validate (0 for a correct run). */
uint64_t oracle_refshape_generate_valid(const oracle_desc* d, uint64_t n_obj, uint64_t* bad_out);

#ifdef __cplusplus
}
#endif
#endif
