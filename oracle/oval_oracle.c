/*
 * oval_oracle.c — CPU restatement of peng225/oval's internal/pattern.
 *
 * TEST INFRASTRUCTURE ONLY (see oval_oracle.h).  Written to follow the
 * reference statement by statement, including its per-unit formatting and
 * append-copy structure, so that it doubles as the "port" CPU baseline.
 * Build with -ffp-contract=off (decide_size must not fuse multiply-adds).
 */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime */
#include "oval_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* unicode/utf8 (Go standard library; not part of the reference tree)        */
/* ------------------------------------------------------------------------ */

#define RUNE_ERROR 0xFFFDu
#define RUNE_SELF  0x80u

/* utf8.first[]: low 3 bits = sequence length, high 4 bits = index into the
 * accept ranges of the second byte; 0xF1 = invalid ("xx"), 0xF0 = ASCII. */
static uint8_t utf8_first(uint8_t c)
{
    if (c < 0x80) return 0xF0;              /* as */
    if (c < 0xC2) return 0xF1;              /* xx: continuation, C0, C1 */
    if (c < 0xE0) return 0x02;              /* s1 */
    if (c == 0xE0) return 0x13;             /* s2 */
    if (c < 0xED) return 0x03;              /* s3 */
    if (c == 0xED) return 0x23;             /* s4 */
    if (c < 0xF0) return 0x03;              /* s3 */
    if (c == 0xF0) return 0x34;             /* s5 */
    if (c < 0xF4) return 0x04;              /* s6 */
    if (c == 0xF4) return 0x44;             /* s7 */
    return 0xF1;                            /* xx */
}

static void utf8_accept(uint8_t idx, uint8_t* lo, uint8_t* hi)
{
    static const uint8_t ranges[5][2] = {
        {0x80, 0xBF}, {0xA0, 0xBF}, {0x80, 0x9F}, {0x90, 0xBF}, {0x80, 0x8F}};
    *lo = ranges[idx][0];
    *hi = ranges[idx][1];
}

/* utf8.DecodeRuneInString */
static size_t utf8_decode(const uint8_t* s, size_t n, uint32_t* r)
{
    if (n < 1) { *r = RUNE_ERROR; return 0; }
    uint8_t s0 = s[0];
    uint8_t x = utf8_first(s0);
    if (x >= 0xF0) {
        *r = (x == 0xF0) ? s0 : RUNE_ERROR;
        return 1;
    }
    size_t sz = x & 7;
    uint8_t lo, hi;
    utf8_accept(x >> 4, &lo, &hi);
    if (n < sz) { *r = RUNE_ERROR; return 1; }
    uint8_t s1 = s[1];
    if (s1 < lo || hi < s1) { *r = RUNE_ERROR; return 1; }
    if (sz <= 2) { *r = ((uint32_t)(s0 & 0x1F) << 6) | (s1 & 0x3F); return 2; }
    uint8_t s2 = s[2];
    if (s2 < 0x80 || 0xBF < s2) { *r = RUNE_ERROR; return 1; }
    if (sz <= 3) {
        *r = ((uint32_t)(s0 & 0x0F) << 12) | ((uint32_t)(s1 & 0x3F) << 6) | (s2 & 0x3F);
        return 3;
    }
    uint8_t s3 = s[3];
    if (s3 < 0x80 || 0xBF < s3) { *r = RUNE_ERROR; return 1; }
    *r = ((uint32_t)(s0 & 0x07) << 18) | ((uint32_t)(s1 & 0x3F) << 12) |
         ((uint32_t)(s2 & 0x3F) << 6) | (s3 & 0x3F);
    return 4;
}

/* utf8.DecodeLastRuneInString */
static size_t utf8_decode_last(const uint8_t* s, size_t n, uint32_t* r)
{
    if (n == 0) { *r = RUNE_ERROR; return 0; }
    long end = (long)n;
    long start = end - 1;
    if (s[start] < RUNE_SELF) { *r = s[start]; return 1; }
    long lim = end - 4;                      /* UTFMax */
    if (lim < 0) lim = 0;
    for (start--; start >= lim; start--) {
        if ((s[start] & 0xC0) != 0x80) break; /* utf8.RuneStart */
    }
    if (start < 0) start = 0;
    size_t size = utf8_decode(s + start, (size_t)(end - start), r);
    if (start + (long)size != end) { *r = RUNE_ERROR; return 1; }
    return size;
}

/* utf8.RuneCountInString: an invalid byte counts as one rune */
size_t oracle_rune_count(const uint8_t* s, size_t n)
{
    size_t cnt = 0;
    for (size_t i = 0; i < n; cnt++) {
        uint32_t r;
        i += utf8_decode(s + i, n - i, &r);
    }
    return cnt;
}

/* unicode.IsSpace */
static int unicode_is_space(uint32_t r)
{
    if (r <= 0xFF) {
        switch (r) {
        case '\t': case '\n': case '\v': case '\f': case '\r': case ' ':
        case 0x85: case 0xA0:
            return 1;
        }
        return 0;
    }
    /* unicode.White_Space, excluding Latin-1 */
    if (r == 0x1680) return 1;
    if (r >= 0x2000 && r <= 0x200A) return 1;
    if (r == 0x2028 || r == 0x2029 || r == 0x202F || r == 0x205F || r == 0x3000) return 1;
    return 0;
}

/* strings.TrimSpace: the result is s[*start:*stop] */
void oracle_trim_space(const uint8_t* s, size_t n, size_t* start_out, size_t* stop_out)
{
    /* strings.TrimLeftFunc(s, unicode.IsSpace); the ASCII fast path of
     * TrimSpace agrees with it byte for byte on ASCII input */
    size_t start = 0;
    while (start < n) {
        uint32_t r;
        size_t w = utf8_decode(s + start, n - start, &r);
        if (!unicode_is_space(r)) break;
        start += w;
    }
    /* strings.TrimRightFunc(s[start:], unicode.IsSpace) */
    size_t stop = n;
    while (stop > start) {
        uint32_t r;
        size_t w = utf8_decode_last(s + start, stop - start, &r);
        if (!unicode_is_space(r)) break;
        stop -= w;
    }
    *start_out = start;
    *stop_out = stop;
}

/* fmt's padString for the verb "%-<width>s" (minus flag: pad on the right);
 * the width counts runes, not bytes */
size_t oracle_pad_field(const uint8_t* s, size_t n, int width, uint8_t* out)
{
    long pad = (long)width - (long)oracle_rune_count(s, n);
    memcpy(out, s, n);
    size_t len = n;
    for (; pad > 0; pad--) out[len++] = ' ';
    return len;
}

/* ------------------------------------------------------------------------ */
/* internal/object/object.go                                                  */
/* ------------------------------------------------------------------------ */

/* object.go:48-50  fmt.Sprintf("%s%010x", KeyShortPrefix, objID) */
size_t oracle_generate_key(int64_t obj_id, char* out)
{
    if (obj_id < 0) /* %x of a negative int64 prints a minus sign and the magnitude */
        return (size_t)sprintf(out, "ov-%09llx", (unsigned long long)(-(obj_id + 1)) + 1ull);
    return (size_t)sprintf(out, "ov%010llx", (unsigned long long)obj_id);
}

/* ------------------------------------------------------------------------ */
/* internal/pattern/pattern.go                                                */
/* ------------------------------------------------------------------------ */

static void put_u32le(uint8_t* p, uint32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static void put_u64le(uint8_t* p, uint64_t v)
{
    put_u32le(p, (uint32_t)v);
    put_u32le(p + 4, (uint32_t)(v >> 32));
}
static uint32_t get_u32le(const uint8_t* p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* pattern.go:94-116 generateDataUnit */
size_t oracle_generate_unit(int64_t unit_count, int64_t worker_id,
                            const uint8_t* bucket, size_t blen,
                            const uint8_t* key, size_t klen,
                            int64_t write_count, uint64_t unix_micro, uint8_t* out)
{
    /* :95,98  Sprintf("%-16s%-12s", bucketName, obj.Key) appended to the unit */
    size_t len = 0;
    len += oracle_pad_field(bucket, blen, ORACLE_MAX_BUCKET, out + len);
    len += oracle_pad_field(key, klen, ORACLE_MAX_KEY, out + len);

    /* :96 */
    int64_t offset_in_object = unit_count * ORACLE_UNIT_SIZE;

    /* :100-107 numBinBuf */
    uint8_t num[ORACLE_HDR_NUM];
    put_u32le(num + 0, (uint32_t)write_count);
    put_u32le(num + 4, (uint32_t)offset_in_object);
    put_u32le(num + 8, (uint32_t)worker_id);
    put_u64le(num + 12, unix_micro);
    memcpy(out + len, num, ORACLE_HDR_NUM);
    len += ORACLE_HDR_NUM;

    /* :109-114 the body: each byte is its own offset; the loop starts at the
     * nominal header size whatever the real header length turned out to be */
    int start = ORACLE_MAX_BUCKET + ORACLE_MAX_KEY + ORACLE_HDR_NUM;
    for (int i = start; i < ORACLE_UNIT_SIZE; i += 4) {
        out[len++] = (uint8_t)i;
        out[len++] = (uint8_t)(i + 1);
        out[len++] = (uint8_t)(i + 2);
        out[len++] = (uint8_t)(i + 3);
    }
    return len;
}

/* pattern.go:73-92 Generate */
int oracle_generate(int64_t data_size, int64_t worker_id,
                    const uint8_t* bucket, size_t blen,
                    const uint8_t* key, size_t klen,
                    int64_t write_count, uint64_t unix_micro,
                    uint8_t* out, char* err, size_t err_cap)
{
    /* :74-76 */
    if (blen > ORACLE_MAX_BUCKET) blen = ORACLE_MAX_BUCKET;

    if (data_size < 0) {
        if (err) snprintf(err, err_cap, "negative data size");  /* make() panics in Go */
        return 1;
    }
    /* :78 data := make([]byte, 0, dataSize); append grows it if units overflow */
    size_t cap = (size_t)data_size + 1;
    size_t len = 0;
    uint8_t* data = (uint8_t*)malloc(cap);
    uint8_t* unit = (uint8_t*)malloc(blen + klen + 64 + ORACLE_UNIT_SIZE);
    if (!data || !unit) { free(data); free(unit); if (err) snprintf(err, err_cap, "out of memory"); return 1; }

    /* :79-85 */
    for (int64_t i = 0; i < data_size / ORACLE_UNIT_SIZE; i++) {
        size_t ulen = oracle_generate_unit(i, worker_id, bucket, blen, key, klen,
                                           write_count, unix_micro, unit);
        if (len + ulen > cap) {
            cap = (len + ulen) * 2;
            uint8_t* nd = (uint8_t*)realloc(data, cap);
            if (!nd) { free(data); free(unit); if (err) snprintf(err, err_cap, "out of memory"); return 1; }
            data = nd;
        }
        memcpy(data + len, unit, ulen);
        len += ulen;
    }
    free(unit);

    /* :87-89 */
    if ((int64_t)len != data_size) {
        if (err)
            snprintf(err, err_cap, "generated data size is wrong. (expected: %lld, actual: %lld)",
                     (long long)data_size, (long long)len);
        free(data);
        return 1;
    }
    memcpy(out, data, len);
    free(data);
    return 0;
}

/* encoding/hex.Dump (Go standard library) */
size_t oracle_hex_dump(const uint8_t* data, size_t n, char* out, size_t cap)
{
    static const char hexd[] = "0123456789abcdef";
    size_t len = 0;
#define PUT(c) do { if (len + 1 < cap) out[len] = (char)(c); len++; } while (0)
    char right[18];
    size_t used = 0;
    for (size_t i = 0; i < n; i++) {
        if (used == 0) {
            uint32_t off = (uint32_t)i;
            for (int sh = 28; sh >= 0; sh -= 4) PUT(hexd[(off >> sh) & 15]);
            PUT(' '); PUT(' ');
        }
        PUT(hexd[data[i] >> 4]); PUT(hexd[data[i] & 15]); PUT(' ');
        if (used == 7) PUT(' ');
        else if (used == 15) { PUT(' '); PUT('|'); }
        right[used] = (data[i] < 32 || data[i] > 126) ? '.' : (char)data[i];
        used++;
        if (used == 16) {
            right[16] = '|'; right[17] = '\n';
            for (int k = 0; k < 18; k++) PUT(right[k]);
            used = 0;
        }
    }
    if (used != 0) {                         /* dumper.Close on a partial line */
        size_t nbytes = used;
        for (; used < 16; used++) {
            PUT(' '); PUT(' '); PUT(' ');
            if (used == 7) PUT(' ');
            else if (used == 15) { PUT(' '); PUT('|'); }
        }
        right[nbytes] = '|'; right[nbytes + 1] = '\n';
        for (size_t k = 0; k < nbytes + 2; k++) PUT(right[k]);
    }
#undef PUT
    if (cap) out[len < cap ? len : cap - 1] = 0;
    return len;
}

/* pattern.go:182-206 dump */
size_t oracle_dump(const char* data, size_t n, char* out, size_t cap)
{
    static const char* explanation[3] = {
        "          ^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^ bucket name\n",
        "          ^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^ key name\n"
        "                                               ^^^^^^^^^^^ write count\n",
        "          ^^^^^^^^^^^ byte offset in this object\n"
        "                      ^^^^^^^^^^^ worker ID\n"
        "                                   ^^^^^^^^^^^^^^^^^^^^^^^ unix time (micro sec)\n",
    };
    const size_t line_size = 79;
    size_t len = 0;
#define APPEND(p, k) do { for (size_t q_ = 0; q_ < (k); q_++) { if (len + 1 < cap) out[len] = (p)[q_]; len++; } } while (0)
    if (n == 0) { if (cap) out[0] = 0; return 0; }        /* :183-185 */
    for (int e = 0; e < 3; e++) {                          /* :194-203 */
        if (n < line_size) {
            APPEND(data, n);
            APPEND(explanation[e], strlen(explanation[e]));
            if (cap) out[len < cap ? len : cap - 1] = 0;
            return len;
        }
        APPEND(data, line_size);
        APPEND(explanation[e], strlen(explanation[e]));
        data += line_size;
        n -= line_size;
    }
    APPEND(data, n);                                       /* :204 */
#undef APPEND
    if (cap) out[len < cap ? len : cap - 1] = 0;
    return len;
}

/* small append-with-format helper for the error text */
static void errcat(char* err, size_t cap, size_t* len, const char* s, size_t n)
{
    if (!err) return;
    for (size_t i = 0; i < n; i++) {
        if (*len + 1 < cap) err[*len] = s[i];
        (*len)++;
    }
    if (cap) err[*len < cap ? *len : cap - 1] = 0;
}
static void errcat_str(char* err, size_t cap, size_t* len, const char* s)
{
    errcat(err, cap, len, s, strlen(s));
}

/* pattern.go:136-180 validDataUnit */
int oracle_valid_unit(int64_t unit_count, int64_t worker_id,
                      const uint8_t* exp_bucket, size_t blen,
                      const uint8_t* key, size_t klen,
                      int64_t write_count, const uint8_t* data,
                      char* err, size_t err_cap)
{
    int mask = 0;
    size_t elen = 0;
    char num[96];
    if (err && err_cap) err[0] = 0;

    /* :137-143 bucket name */
    size_t current = ORACLE_MAX_BUCKET;
    size_t a, b;
    oracle_trim_space(data, ORACLE_MAX_BUCKET, &a, &b);
    if (!(b - a == blen && memcmp(data + a, exp_bucket, blen) == 0)) {
        mask |= 1;
        errcat_str(err, err_cap, &elen, "- Bucket name is wrong. (expected = \"");
        errcat(err, err_cap, &elen, (const char*)exp_bucket, blen);
        errcat_str(err, err_cap, &elen, "\", actual = \"");
        errcat(err, err_cap, &elen, (const char*)data + a, b - a);
        errcat_str(err, err_cap, &elen, "\")\n");
    }

    /* :145-150 key */
    const uint8_t* kf = data + current;
    current += ORACLE_MAX_KEY;
    oracle_trim_space(kf, ORACLE_MAX_KEY, &a, &b);
    if (!(b - a == klen && memcmp(kf + a, key, klen) == 0)) {
        mask |= 2;
        errcat_str(err, err_cap, &elen, "- Key name is wrong. (expected = \"");
        errcat(err, err_cap, &elen, (const char*)key, klen);
        errcat_str(err, err_cap, &elen, "\", actual = \"");
        errcat(err, err_cap, &elen, (const char*)kf + a, b - a);
        errcat_str(err, err_cap, &elen, "\")\n");
    }

    /* :152-157 write count */
    uint32_t wc = get_u32le(data + current);
    current += 4;
    if ((uint32_t)write_count != wc) {
        mask |= 4;
        snprintf(num, sizeof num, "- WriteCount is wrong. (expected = \"%lld\", actual = \"%u\")\n",
                 (long long)write_count, wc);
        errcat_str(err, err_cap, &elen, num);
    }

    /* :159-164 offset in object */
    uint32_t off = get_u32le(data + current);
    current += 4;
    int64_t exp_off = unit_count * ORACLE_UNIT_SIZE;
    if ((uint32_t)exp_off != off) {
        mask |= 8;
        snprintf(num, sizeof num, "- OffsetInObject is wrong. (expected = \"%lld\", actual = \"%u\")\n",
                 (long long)exp_off, off);
        errcat_str(err, err_cap, &elen, num);
    }

    /* :166-170 worker ID, compared as int */
    int64_t actual_worker = (int64_t)get_u32le(data + current);
    if (worker_id != actual_worker) {
        mask |= 16;
        snprintf(num, sizeof num, "- WorkerID is wrong. (expected = \"%lld\", actual = \"%lld\")\n",
                 (long long)worker_id, (long long)actual_worker);
        errcat_str(err, err_cap, &elen, num);
    }

    /* :172 the unix time area is skipped; :174-177 */
    if (mask != 0 && err) {
        char hex[79 * 16 + 8];
        char dmp[79 * 16 + 512];
        size_t hl = oracle_hex_dump(data, ORACLE_UNIT_SIZE, hex, sizeof hex);
        size_t dl = oracle_dump(hex, hl, dmp, sizeof dmp);
        errcat(err, err_cap, &elen, dmp, dl);
    }
    return mask;
}

/* pattern.go:118-134 Valid; the io.Reader is an in-memory body */
int oracle_valid(int64_t worker_id, const uint8_t* exp_bucket, size_t blen,
                 const uint8_t* key, size_t klen, int64_t size, int64_t write_count,
                 const uint8_t* body, size_t body_len,
                 int64_t* bad_unit, char* err, size_t err_cap)
{
    /* :119-121 */
    if (blen > ORACLE_MAX_BUCKET) blen = ORACLE_MAX_BUCKET;
    uint8_t data[ORACLE_UNIT_SIZE];
    size_t pos = 0;
    if (err && err_cap) err[0] = 0;
    for (int64_t i = 0; i < size / ORACLE_UNIT_SIZE; i++) {
        /* :124 io.ReadFull(reader, data) */
        size_t n = body_len - pos;
        if (n > ORACLE_UNIT_SIZE) n = ORACLE_UNIT_SIZE;
        memcpy(data, body + pos, n);
        pos += n;
        if (n < ORACLE_UNIT_SIZE) {
            /* :125-127 */
            if (bad_unit) *bad_unit = i;
            if (err) {
                char hex[79 * 16 + 8];
                char dmp[79 * 16 + 512];
                size_t elen = 0;
                char head[96];
                snprintf(head, sizeof head,
                         "could not read some data. (expected: %dbyte, actual: %zubyte)\n",
                         ORACLE_UNIT_SIZE, n);
                errcat_str(err, err_cap, &elen, head);
                size_t hl = oracle_hex_dump(data, n, hex, sizeof hex);
                size_t dl = oracle_dump(hex, hl, dmp, sizeof dmp);
                errcat(err, err_cap, &elen, dmp, dl);
            }
            return 2;
        }
        /* :128-131 */
        if (oracle_valid_unit(i, worker_id, exp_bucket, blen, key, klen, write_count,
                              data, err, err_cap) != 0) {
            if (bad_unit) *bad_unit = i;
            return 1;
        }
    }
    return 0;
}

/* ---- math.Log / math.Log2 of the Go standard library (pure-Go versions,
 * themselves FreeBSD's e_log.c) so that decide_size rounds like Go does ---- */
static double go_log(double x)
{
    const double Ln2Hi = 6.93147180369123816490e-01;
    const double Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01;
    const double L2 = 3.999999999940941908e-01;
    const double L3 = 2.857142874366239149e-01;
    const double L4 = 2.222219843214978396e-01;
    const double L5 = 1.818357216161805012e-01;
    const double L6 = 1.531383769920937332e-01;
    const double L7 = 1.479819860511658591e-01;
    if (isnan(x) || isinf(x)) return x > 0 ? x : NAN;
    if (x < 0) return NAN;
    if (x == 0) return -INFINITY;
    int ki;
    double f1 = frexp(x, &ki);
    if (f1 < 1.41421356237309504880168872420969808 / 2) { f1 *= 2; ki--; }
    double f = f1 - 1;
    double k = (double)ki;
    double s = f / (2 + f);
    double s2 = s * s;
    double s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

static double go_log2(double x)
{
    int e;
    double frac = frexp(x, &e);
    if (frac == 0.5) return (double)(e - 1);
    return go_log(frac) * (1 / 0.693147180559945309417232121458176568) + (double)e;
}

/* pattern.go:22-71 DecideSize; y stands for rand.Float64() */
int oracle_decide_size(int64_t min_size, int64_t max_size, double y,
                       int64_t* size_out, char* err, size_t err_cap)
{
    if (min_size < ORACLE_UNIT_SIZE) {
        if (err) snprintf(err, err_cap, "minSize should be larger than or equal to %d", ORACLE_UNIT_SIZE);
        return 1;
    }
    if (min_size % ORACLE_UNIT_SIZE != 0) {
        if (err) snprintf(err, err_cap, "minSize should be a multiple of %d", ORACLE_UNIT_SIZE);
        return 1;
    }
    if (max_size % ORACLE_UNIT_SIZE != 0) {
        if (err) snprintf(err, err_cap, "maxSize should be a multiple of %d", ORACLE_UNIT_SIZE);
        return 1;
    }
    if (max_size < min_size) {
        if (err) snprintf(err, err_cap, "maxSize should be larger than or equal to minSize");
        return 1;
    }
    double x = -go_log2(1 - 0.5 * y);                       /* :64 */
    double scaled = (double)((max_size - min_size) / ORACLE_UNIT_SIZE + 1) * x;
    *size_out = min_size + ORACLE_UNIT_SIZE * (int64_t)scaled;  /* :70 */
    return 0;
}

/* ------------------------------------------------------------------------ */
/* batch forms                                                                */
/* ------------------------------------------------------------------------ */

void oracle_fill_batch(const oracle_desc* d, uint64_t n_obj, uint8_t* body)
{
    uint8_t unit[ORACLE_UNIT_SIZE + 64];
    for (uint64_t o = 0; o < n_obj; o++) {
        const oracle_desc* e = &d[o];
        for (uint32_t u = 0; u < e->n_units; u++) {
            size_t len = oracle_generate_unit((int64_t)e->first_unit + u, (int64_t)e->worker_id,
                                              e->bucket, e->bucket_len, e->key, e->key_len,
                                              (int64_t)e->write_count, e->unix_micro, unit);
            if (len != ORACLE_UNIT_SIZE) abort();   /* descriptors hold 28-byte fields only */
            memcpy(body + e->body_offset + (uint64_t)u * ORACLE_UNIT_SIZE, unit, ORACLE_UNIT_SIZE);
        }
    }
}

void oracle_validate_batch(const oracle_desc* d, uint64_t n_obj,
                           const uint8_t* body, uint64_t body_bytes,
                           oracle_result* res, uint64_t* counters)
{
    uint64_t c[ORACLE_NUM_COUNTERS];
    memset(c, 0, sizeof c);
    for (uint64_t o = 0; o < n_obj; o++) {
        const oracle_desc* e = &d[o];
        oracle_result r = {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0xFFFFFFFFu};
        int64_t worker = (e->flags & 4) ? -1 : (int64_t)e->worker_id; /* out-of-range worker IDs never match */
        for (uint32_t u = 0; u < e->n_units; u++) {
            uint64_t off = e->body_offset + (uint64_t)u * ORACLE_UNIT_SIZE;
            if (off + ORACLE_UNIT_SIZE > body_bytes) break;
            const uint8_t* p = body + off;
            uint32_t unit = e->first_unit + u;
            int m = oracle_valid_unit((int64_t)unit, worker, e->bucket, e->bucket_len,
                                      e->key, e->key_len, (int64_t)e->write_count, p, NULL, 0);
            int filler_bad = 0;
            for (int i = 48; i < ORACLE_UNIT_SIZE; i++)
                if (p[i] != (uint8_t)i) { filler_bad = 1; break; }
            c[2]++;
            if (m) {
                if (r.first_bad_unit == 0xFFFFFFFFu) { r.first_bad_unit = unit; r.first_bad_fields = (uint32_t)m; }
                r.bad_units++;
                c[3]++;
                if (m & 1) c[4]++;
                if (m & 2) c[5]++;
                if (m & 4) c[6]++;
                if (m & 8) c[7]++;
                if (m & 16) c[8]++;
            }
            if (m || filler_bad) {
                if (r.strict_first_bad_unit == 0xFFFFFFFFu) r.strict_first_bad_unit = unit;
                c[9]++;
            }
        }
        c[0]++;
        if (r.bad_units) c[1]++;
        res[o] = r;
    }
    if (counters)
        for (int i = 0; i < ORACLE_NUM_COUNTERS; i++) counters[i] += c[i];
}

/* ------------------------------------------------------------------------ */
/* timing variant shaped like the reference's code                           */
/* ------------------------------------------------------------------------ */
/* For the CPU baseline only: Generate then Valid of every object of the table,
 * paying per unit what pattern.go pays per unit -- the format string rebuilt
 * and the names formatted through a printf (pattern.go:95,98), two string to
 * []byte conversions and two buffers allocated (:97-100), the clock read
 * (:103), 52 four-byte appends (:109-114), the unit appended to the object's
 * buffer (:84); and on the way back a 256-byte read into a buffer (:124), two
 * []byte to string conversions (:139,146), TrimSpace and the compares
 * (:140-170).  The time stamp is the wall clock, so its bytes differ from a
 * table's pinned time; nothing validates them.  Returns the number of units
 * generated and validated; *bad_out counts units that did not validate (0 for
 * a correct run). */
static uint8_t* refshape_unit(int64_t unit_count, int64_t worker_id, const char* bucket, const char* key,
                              int64_t write_count, size_t* len_out)
{
    char format[32];
    snprintf(format, sizeof format, "%%-%ds%%-%ds", ORACLE_MAX_BUCKET, ORACLE_MAX_KEY);   /* :95 */
    int64_t offset_in_object = unit_count * ORACLE_UNIT_SIZE;                              /* :96 */
    uint8_t* data = (uint8_t*)malloc(ORACLE_UNIT_SIZE);                                    /* :97 */
    size_t len = 0;
    char* header = (char*)malloc(ORACLE_MAX_BUCKET + ORACLE_MAX_KEY + 64);                 /* Sprintf's string */
    int hl = snprintf(header, ORACLE_MAX_BUCKET + ORACLE_MAX_KEY + 64, format, bucket, key);
    uint8_t* hbytes = (uint8_t*)malloc((size_t)hl + 1);                                    /* []byte(string) */
    memcpy(hbytes, header, (size_t)hl);
    memcpy(data + len, hbytes, (size_t)hl);                                                /* :98 append */
    len += (size_t)hl;
    uint8_t* num = (uint8_t*)malloc(ORACLE_HDR_NUM);                                       /* :100 */
    put_u32le(num + 0, (uint32_t)write_count);
    put_u32le(num + 4, (uint32_t)offset_in_object);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);                                                    /* :103 time.Now() */
    uint64_t micro = (uint64_t)ts.tv_sec * 1000000ull + (uint64_t)ts.tv_nsec / 1000ull;
    put_u32le(num + 8, (uint32_t)worker_id);
    put_u64le(num + 12, micro);
    memcpy(data + len, num, ORACLE_HDR_NUM);                                               /* :107 */
    len += ORACLE_HDR_NUM;
    for (int i = ORACLE_MAX_BUCKET + ORACLE_MAX_KEY + ORACLE_HDR_NUM; i < ORACLE_UNIT_SIZE; i += 4) {
        uint8_t four[4] = {(uint8_t)i, (uint8_t)(i + 1), (uint8_t)(i + 2), (uint8_t)(i + 3)};
        memcpy(data + len, four, 4);                                                       /* :113 append */
        len += 4;
    }
    free(num);
    free(hbytes);
    free(header);
    *len_out = len;
    return data;
}

uint64_t oracle_refshape_generate_valid(const oracle_desc* d, uint64_t n_obj, uint64_t* bad_out)
{
    uint64_t units = 0, bad = 0;
    for (uint64_t o = 0; o < n_obj; o++) {
        const oracle_desc* e = &d[o];
        char bucket[ORACLE_MAX_BUCKET + 1], key[ORACLE_MAX_KEY + 1];
        memcpy(bucket, e->bucket, e->bucket_len); bucket[e->bucket_len] = 0;
        memcpy(key, e->key, e->key_len); key[e->key_len] = 0;
        /* Generate, pattern.go:73-92 */
        size_t data_size = (size_t)e->n_units * ORACLE_UNIT_SIZE, len = 0;
        uint8_t* data = (uint8_t*)malloc(data_size ? data_size : 1);                       /* :78 */
        for (uint32_t u = 0; u < e->n_units; u++) {
            size_t ulen = 0;
            uint8_t* unit = refshape_unit((int64_t)e->first_unit + u, (int64_t)e->worker_id, bucket, key,
                                          (int64_t)e->write_count, &ulen);
            if (len + ulen <= data_size) memcpy(data + len, unit, ulen);                   /* :84 append */
            len += ulen;
            free(unit);
        }
        if (len != data_size) bad += e->n_units;                                           /* :87-89 */
        /* Valid, pattern.go:118-134 over a reader on those bytes */
        uint8_t buf[ORACLE_UNIT_SIZE];                                                     /* :122 */
        for (uint32_t u = 0; u < e->n_units && len == data_size; u++) {
            memcpy(buf, data + (size_t)u * ORACLE_UNIT_SIZE, ORACLE_UNIT_SIZE);            /* :124 io.ReadFull */
            char* s1 = (char*)malloc(ORACLE_MAX_BUCKET);                                   /* :139 string(data[0:16]) */
            memcpy(s1, buf, ORACLE_MAX_BUCKET);
            char* s2 = (char*)malloc(ORACLE_MAX_KEY);                                      /* :146 */
            memcpy(s2, buf + ORACLE_MAX_BUCKET, ORACLE_MAX_KEY);
            int m = oracle_valid_unit((int64_t)e->first_unit + u, (int64_t)e->worker_id, e->bucket, e->bucket_len,
                                      e->key, e->key_len, (int64_t)e->write_count, buf, NULL, 0);
            if (m) bad++;
            free(s2);
            free(s1);
        }
        units += e->n_units;
        free(data);
    }
    if (bad_out) *bad_out = bad;
    return units;
}
