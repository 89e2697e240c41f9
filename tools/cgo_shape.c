/*
 * cgo_shape.c — a plain-C consumer of liboval_hip.so that calls it the way the
 * cgo files in go/internal/pattern do (pattern_hip.go, parts_hip.go,
 * batch_hip.go), for a box without a Go toolchain: names as pointer + length
 * with no terminating NUL (Go strings), outputs into caller-owned malloc memory
 * (Go slices), error texts into malloc'ed buffers of errTextCap = 8192 bytes,
 * the io.Reader behind Valid as a callback with a `user` pointer, the batch
 * forms with C arrays of pointers, all from 64 threads at once (cgo runs every
 * call on the calling goroutine's OS thread).
 *
 * The reference surface this stands in front of: pattern.DecideSize
 * (internal/pattern/pattern.go:22), Generate (:73), generateDataUnit (:94),
 * Valid (:118), validDataUnit (:136); the multipart slicing of
 * internal/s3client/s3client.go:164-236.
 *
 * Built with gcc -std=c11 (not hipcc) against include/ only:
 *   gcc -std=c11 -O2 -Wall -Wextra -pedantic -Iinclude tools/cgo_shape.c -o build/cgo_shape \
 *       -Loval_amd -loval_hip -lpthread -Wl,-rpath,'$ORIGIN/../oval_amd' -Wl,-rpath-link,/opt/rocm/lib
 *
 *   cgo_shape [threads (64)] [rounds per thread (6)]
 * exits 0 and prints one summary line when every check held.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oval_hip.h"
#include "oval_pattern.h"

#define ERR_TEXT_CAP 8192        /* errTextCap of pattern_hip.go, batchErrStride of batch_hip.go */
#define READ_MAX 1000003u        /* the reader hands out at most this many bytes per call */
#define T0 1709558129112997ll

static atomic_int g_failures;
static atomic_long g_calls, g_foreign_thread_reads, g_reads;

#define CHECK(cond, ...)                                                     \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "cgo_shape: %s:%d: ", __FILE__, __LINE__);       \
            fprintf(stderr, __VA_ARGS__);                                    \
            fprintf(stderr, "\n");                                           \
            atomic_fetch_add(&g_failures, 1);                                \
            goto out;                                                        \
        }                                                                    \
    } while (0)

/* a Go string: bytes + length, not NUL-terminated (the byte after it is poison) */
typedef struct { char* p; size_t n; } gostr;

static gostr gostr_of(const char* s)
{
    gostr g;
    g.n = strlen(s);
    g.p = malloc(g.n + 1);
    memcpy(g.p, s, g.n);
    g.p[g.n] = '#';
    return g;
}

/* the readerBox of pattern_hip.go: what `user` points at */
typedef struct {
    const uint8_t* body;
    size_t len, pos;
    pthread_t caller;            /* the thread that called oval_pattern_valid_stream */
} reader_box;

/* ovalGoRead: Reader.Read into p, 0 at EOF */
static size_t shape_read(void* user, uint8_t* p, size_t n)
{
    reader_box* r = user;
    atomic_fetch_add(&g_reads, 1);
    if (!pthread_equal(pthread_self(), r->caller)) atomic_fetch_add(&g_foreign_thread_reads, 1);
    size_t take = r->len - r->pos;
    if (take > n) take = n;
    if (take > READ_MAX) take = READ_MAX;
    memcpy(p, r->body + r->pos, take);
    r->pos += take;
    return take;
}

static int valid_through_reader(int64_t wid, gostr b, gostr k, int64_t size, int64_t wc, const uint8_t* body, size_t len,
                                char* text)
{
    reader_box* box = malloc(sizeof *box);   /* C memory, as the cgo.Handle cell is */
    box->body = body; box->len = len; box->pos = 0; box->caller = pthread_self();
    text[0] = 0;
    const int rc = oval_pattern_valid_stream(wid, b.p, b.n, k.p, k.n, size, wc, shape_read, box, text, ERR_TEXT_CAP);
    free(box);
    atomic_fetch_add(&g_calls, 1);
    return rc;
}

/* body == generateDataUnit(u) for every unit u (pattern.go:79-85), through the library's own single-unit call */
static int units_equal(const uint8_t* body, int64_t size, int64_t wid, gostr b, gostr k, int64_t wc, char* text, int64_t step)
{
    uint8_t unit[256];
    for (int64_t u = 0; u < size / 256; u += step) {
        const int rc = oval_pattern_generate_data_unit(u, wid, b.p, b.n, k.p, k.n, wc, unit, text, ERR_TEXT_CAP);
        atomic_fetch_add(&g_calls, 1);
        if (rc != 0 || memcmp(unit, body + u * 256, 256) != 0) return 0;
    }
    return 1;
}

typedef struct { int index, rounds; } worker_arg;

static void* worker(void* argp)
{
    const worker_arg* arg = argp;
    const int t = arg->index;
    char* text = malloc(ERR_TEXT_CAP);
    char* small = malloc(256);
    uint8_t* data = NULL;
    gostr bucket = gostr_of(t % 3 ? "test-bucket" : "test-long-long-long-bucket");   /* cut to 16 bytes by the library */
    /* generateDataUnit takes the name as Generate hands it on: already cut to 16 bytes (pattern.go:74-76) */
    gostr bucket16 = gostr_of(t % 3 ? "test-bucket" : "test-long-long-l");
    char keybuf[32];
    const int64_t wid = 0x561B + t;

    for (int round = 0; round < arg->rounds; ++round) {
        snprintf(keybuf, sizeof keybuf, "ov%010llx", (unsigned long long)(((uint64_t)t << 24) + (uint64_t)round));
        gostr key = gostr_of(keybuf);
        const int64_t wc = 1 + round;
        /* DecideSize -> Generate -> Valid, as Worker.Put does (internal/runner/worker.go:77-109) */
        int64_t size = 0;
        int rc = oval_pattern_decide_size(4096, 16384, &size, small, 256);
        CHECK(rc == 0 && size >= 4096 && size <= 16384 + 256 && size % 256 == 0, "decide_size rc %d size %lld", rc, (long long)size);
        free(data);
        data = malloc((size_t)size);
        memset(data, 0xEE, (size_t)size);
        rc = oval_pattern_generate(size, wid, bucket.p, bucket.n, key.p, key.n, wc, data, text, ERR_TEXT_CAP);
        atomic_fetch_add(&g_calls, 1);
        CHECK(rc == 0, "generate rc %d: %s", rc, rc < 0 ? oval_last_error() : text);
        CHECK(units_equal(data, size, wid, bucket16, key, wc, text, round == 0 ? 1 : 7), "generated body differs from generateDataUnit");
        rc = valid_through_reader(wid, bucket, key, size, wc, data, (size_t)size, text);
        CHECK(rc == 0, "valid rc %d: %s", rc, text);
        /* a stale generation in unit 3: exactly the reference's words */
        data[3 * 256 + 28] ^= 1;
        rc = valid_through_reader(wid, bucket, key, size, wc, data, (size_t)size, text);
        CHECK(rc == 1 && strncmp(text, "- WriteCount is wrong. (expected = \"", 36) == 0 && strstr(text, "^^^ bucket name\n"),
              "damaged body: rc %d text %.80s", rc, text);
        data[3 * 256 + 28] ^= 1;
        /* a body that ends early: the short-read message (pattern.go:126) */
        rc = valid_through_reader(wid, bucket, key, size, wc, data, (size_t)size - 100, text);
        CHECK(rc == 1 && strncmp(text, "could not read some data. (expected: 256byte, actual: 156byte)\n", 63) == 0,
              "short body: rc %d text %.80s", rc, text);
        /* the same put-cycle in two steps, as twostep_hip.go does it: begin, yield between polls, collect */
        {
            uint8_t* again = malloc((size_t)size);
            oval_pending* h = NULL;
            rc = oval_pattern_generate_begin(size, wid, bucket.p, bucket.n, key.p, key.n, wc, again, &h);
            if (rc == 0)
                while ((rc = oval_pattern_poll(h, text, ERR_TEXT_CAP)) == OVAL_PATTERN_PENDING) sched_yield();
            const int same = rc == 0 && memcmp(again, data, (size_t)size) == 0;
            again[5 * 256 + 36] ^= 0x20;                         /* worker ID of unit 5 */
            int rc2 = oval_pattern_valid_begin(wid, bucket.p, bucket.n, key.p, key.n, size, wc, again, (size_t)size, &h);
            if (rc2 == 0)
                while ((rc2 = oval_pattern_poll(h, text, ERR_TEXT_CAP)) == OVAL_PATTERN_PENDING) sched_yield();
            free(again);
            atomic_fetch_add(&g_calls, 2);
            CHECK(same, "two-step generate: rc %d, body %s", rc, rc == 0 ? "differs" : "missing");
            CHECK(rc2 == 1 && strncmp(text, "- WorkerID is wrong.", 20) == 0, "two-step valid: rc %d text %.60s", rc2, text);
        }
        /* validDataUnit on one unit of it */
        rc = oval_pattern_valid_data_unit(2, wid, bucket16.p, bucket16.n, key.p, key.n, wc, data + 512, text, ERR_TEXT_CAP);
        CHECK(rc == 0, "valid_data_unit rc %d: %s", rc, text);
        free(key.p);
    }

    /* a body uploaded in parts (s3client.go:164-236): 12 MiB + 512 bytes in 5 MiB parts, against Generate's bytes */
    if (t % 8 == 0) {
        const int64_t size = (12 << 20) + 512, part = 5 << 20;
        gostr key = gostr_of("ov00000000aa");
        free(data);
        data = malloc((size_t)size);
        int rc = oval_pattern_generate(size, wid, bucket.p, bucket.n, key.p, key.n, 9, data, text, ERR_TEXT_CAP);
        CHECK(rc == 0, "large generate rc %d: %s", rc, rc < 0 ? oval_last_error() : text);
        CHECK(units_equal(data, size, wid, bucket16, key, 9, text, 4099), "large body differs from generateDataUnit");
        oval_parts* parts = NULL;
        rc = oval_pattern_generate_parts(size, wid, bucket.p, bucket.n, key.p, key.n, 9, part, &parts, text, ERR_TEXT_CAP);
        CHECK(rc == 0 && parts && oval_parts_count(parts) == 3, "generate_parts rc %d count %lld", rc,
              parts ? (long long)oval_parts_count(parts) : -1ll);
        int same = 1;
        for (int64_t i = 2; i >= 0; --i) {                       /* any order */
            const uint8_t* p = NULL;
            size_t n = 0;
            rc = oval_parts_get(parts, i, &p, &n, text, ERR_TEXT_CAP);
            const size_t want = i < 2 ? (size_t)part : (size_t)(size - 2 * part);
            same &= rc == 0 && n == want && memcmp(p, data + i * part, n) == 0;
            (void)oval_parts_release(parts, i);
        }
        oval_parts_close(parts);
        atomic_fetch_add(&g_calls, 6);
        CHECK(same, "parts differ from the generated body");
        /* and back through a reader that dribbles it: the body is larger than the library's read-ahead chunk */
        oval_pattern_set_stream_chunk(4u << 20);
        rc = valid_through_reader(wid, bucket, key, size, 9, data, (size_t)size, text);
        CHECK(rc == 0, "large valid rc %d: %s", rc, text);
        data[(7 << 20) + 36] ^= 0x10;                            /* worker ID of a unit in the second chunk */
        rc = valid_through_reader(wid, bucket, key, size, 9, data, (size_t)size, text);
        CHECK(rc == 1 && strncmp(text, "- WorkerID is wrong.", 20) == 0, "large damaged: rc %d text %.60s", rc, text);
        free(key.p);
    }

    /* the batch forms, arguments laid out as batch_hip.go lays them out: calloc'ed arrays of pointers and lengths */
    if (t % 4 == 1) {
        enum { N = 48 };
        const char** buckets = calloc(N, sizeof *buckets);
        const char** keys = calloc(N, sizeof *keys);
        size_t* blens = calloc(N, sizeof *blens);
        size_t* klens = calloc(N, sizeof *klens);
        int64_t* wids = calloc(N, 8), *wcs = calloc(N, 8), *sizes = calloc(N, 8);
        int* rcs = calloc(N, sizeof *rcs);
        char* errs = calloc(N, ERR_TEXT_CAP);
        uint8_t** outs = calloc(N, sizeof *outs);
        size_t* lens = calloc(N, sizeof *lens);
        gostr ks[N];
        for (int i = 0; i < N; ++i) {
            snprintf(keybuf, sizeof keybuf, "ov%010llx", (unsigned long long)(((uint64_t)t << 24) + 0x800000u + (uint64_t)i));
            ks[i] = gostr_of(keybuf);
            buckets[i] = bucket.p; blens[i] = bucket.n;
            keys[i] = ks[i].p; klens[i] = ks[i].n;
            wids[i] = wid + i; wcs[i] = 3 + i; sizes[i] = 256 * (1 + (i * 7) % 64);
            outs[i] = malloc((size_t)sizes[i]);
            lens[i] = (size_t)sizes[i];
        }
        int worst = oval_pattern_generate_batch(N, sizes, wids, buckets, blens, keys, klens, wcs, outs, rcs, errs, ERR_TEXT_CAP);
        int ok = worst == 0;
        for (int i = 0; i < N && ok; ++i)
            ok = rcs[i] == 0 && units_equal(outs[i], sizes[i], wids[i], bucket16, ks[i], wcs[i], text, 5);
        outs[5][28] ^= 2;                                        /* item 5: a stale write count in unit 0 */
        lens[9] -= 256;                                          /* item 9: its last unit never arrives */
        if (ok) {
            worst = oval_pattern_valid_batch(N, wids, buckets, blens, keys, klens, sizes, wcs, (const uint8_t* const*)outs, lens, rcs,
                                             errs, ERR_TEXT_CAP);
            ok = worst == 0;
            for (int i = 0; i < N && ok; ++i) {
                const char* e = errs + (size_t)i * ERR_TEXT_CAP;
                if (i == 5) ok = rcs[i] == 1 && strncmp(e, "- WriteCount is wrong.", 22) == 0;
                else if (i == 9) ok = rcs[i] == 1 && strncmp(e, "could not read some data. (expected: 256byte, actual: 0byte)\n", 61) == 0;
                else ok = rcs[i] == 0;
            }
        }
        atomic_fetch_add(&g_calls, 2);
        for (int i = 0; i < N; ++i) { free(outs[i]); free(ks[i].p); }
        free(buckets); free(keys); free(blens); free(klens); free(wids); free(wcs); free(sizes); free(rcs); free(errs);
        free(outs); free(lens);
        CHECK(ok, "batch forms: status %d", worst);
    }
out:
    free(data);
    free(text);
    free(small);
    free(bucket.p);
    free(bucket16.p);
    return NULL;
}

int main(int argc, char** argv)
{
    const int n_threads = argc > 1 ? atoi(argv[1]) : 64;
    const int rounds = argc > 2 ? atoi(argv[2]) : 6;
    if (n_threads < 1 || n_threads > 1024 || rounds < 1) {
        fprintf(stderr, "usage: cgo_shape [threads] [rounds]\n");
        return 2;
    }
    /* what init() of pattern_hip.go does */
    int rc = oval_pattern_init(0);
    if (rc != 0) {
        fprintf(stderr, "cgo_shape: oval_pattern_init: status %d: %s\n", rc, oval_last_error());
        return 1;
    }
    oval_pattern_set_batching(1, 0, 0);
    oval_pattern_set_clock(T0);              /* pinned, so that bodies can be compared with generateDataUnit byte for byte */
    oval_pattern_seed(7);

    pthread_t* th = calloc((size_t)n_threads, sizeof *th);
    worker_arg* args = calloc((size_t)n_threads, sizeof *args);
    for (int t = 0; t < n_threads; ++t) {
        args[t].index = t;
        args[t].rounds = rounds;
        if (pthread_create(&th[t], NULL, worker, &args[t]) != 0) {
            fprintf(stderr, "cgo_shape: pthread_create failed\n");
            return 1;
        }
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    uint64_t batches = 0, jobs = 0;
    oval_pattern_batching_stats(&batches, &jobs);
    oval_pattern_shutdown();                 /* Shutdown() of pattern_hip.go */
    free(th);
    free(args);
    const int failures = atomic_load(&g_failures);
    printf("cgo_shape: %d threads x %d rounds, %ld calls, %ld reader callbacks (%ld on a thread other than the caller's), "
           "%llu launches for %llu device jobs, %d failures\n",
           n_threads, rounds, atomic_load(&g_calls), atomic_load(&g_reads), atomic_load(&g_foreign_thread_reads),
           (unsigned long long)batches, (unsigned long long)jobs, failures);
    return failures == 0 && atomic_load(&g_foreign_thread_reads) == 0 ? 0 : 1;
}
