// oval_kernels.hip — gfx950 (CDNA4, wave64) kernels of the data-pattern engine.
//
// A 256-byte data unit (internal/pattern/pattern.go:94-116) is 16 lanes x 16
// bytes, so one wavefront instruction moves four units with 128-bit accesses
// and each 16-lane group reads or writes two whole 128-byte lines.  Lane j of
// a group owns bytes [16j, 16j+16) of its unit:
//   lane 0      bucket name field                    (pattern.go:98)
//   lane 1      key field (12 B) + write count       (pattern.go:98,101)
//   lane 2      offset in object, worker ID, time    (pattern.go:102-106)
//   lanes 3-15  filler, byte value == byte offset    (pattern.go:109-114)
// Lanes 0..2 keep their slice of the object's 48-byte header template in
// registers, lanes 3..15 keep compile-time constants; nothing but the body
// itself is streamed from or to HBM.
//
// Each 16-lane group walks one chunk of 64 consecutive units, so the four
// groups of a wavefront are free to sit in different objects; a group moves to
// the next descriptor when its unit index passes the end of the current one.
#include "oval_kernels.h"
#include "oval_trimspace.h"

namespace oval {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kNone = 0xFFFFFFFFu;

// bytes 4k..4k+3 of the filler, as a little-endian dword
__device__ __forceinline__ uint32_t filler_dword(uint32_t k) { return 0x03020100u + 0x04040404u * k; }

__device__ __forceinline__ uint32_t reg_byte(const u32x4& v, int i)
{
    uint32_t w = (i & 8) ? ((i & 4) ? v.w : v.z) : ((i & 4) ? v.y : v.x);
    return (w >> ((i & 3) * 8)) & 0xFFu;
}

// What one lane knows about the object its group is currently in.
struct ObjState {
    uint32_t o;      // object index; n_obj when the group has run out of objects
    uint32_t start;  // global unit index of the object's first unit
    uint32_t end;    // one past its last unit (kNone: no object)
    uint32_t ubias;  // lane 2 only: first_unit - start, so unitCount = g + ubias
    uint32_t lf;     // bucket_len | key_len << 8 | flags << 16
    uint32_t force;  // non-zero: this lane's name/worker field needs the exact check
    u32x4    tmpl;   // expected 16 bytes of this lane (lane 2: .x = ubias << 8)
};

__device__ __forceinline__ void span_of(const u32x4* __restrict__ descs, uint32_t o,
                                        uint32_t& start, uint32_t& end, uint32_t& lf)
{
    const u32x4 s3 = descs[4u * o + 3u];          // body_offset (8) | n_units (4) | lens+flags (4)
    start = (s3.x >> 8) | (s3.y << 24);           // body_offset / 256
    end = start + s3.z;
    lf = s3.w;
}

__device__ __forceinline__ void load_template(const u32x4* __restrict__ descs, int j,
                                              const u32x4& cpat, ObjState& st)
{
    st.tmpl = cpat;
    st.force = 0;
    st.ubias = 0;
    if (j < 3) {
        st.tmpl = descs[4u * st.o + (uint32_t)j];
        const uint32_t flags = st.lf >> 16;
        st.force = flags & (1u << j);             // BUCKET_SLOW, KEY_SLOW, WORKER_OOR are bits 0,1,2
        if (j == 2) {
            st.ubias = st.tmpl.x - st.start;      // tmpl.x holds first_unit in the descriptor
            st.tmpl.x = st.ubias << 8;            // + (g << 8) gives uint32(unitCount * 256)
        }
    }
}

__device__ __forceinline__ void enter_object(const u32x4* __restrict__ descs, uint32_t n_obj,
                                             uint32_t o, int j, const u32x4& cpat, ObjState& st)
{
    st.o = o;
    if (o < n_obj) {
        span_of(descs, o, st.start, st.end, st.lf);
        load_template(descs, j, cpat, st);
    } else {
        st.o = n_obj;
        st.start = kNone; st.end = kNone; st.lf = 0; st.force = 0; st.ubias = 0;
        st.tmpl = cpat;
    }
}

// Move to the first later object whose end lies beyond unit g (zero-unit
// objects are stepped over).
__device__ __forceinline__ void advance_object(const u32x4* __restrict__ descs, uint32_t n_obj,
                                               uint32_t g, int j, const u32x4& cpat, ObjState& st)
{
    uint32_t o = st.o;
    for (;;) {
        ++o;
        if (o >= n_obj) { enter_object(descs, n_obj, n_obj, j, cpat, st); return; }
        span_of(descs, o, st.start, st.end, st.lf);
        if (g < st.end) break;
    }
    st.o = o;
    load_template(descs, j, cpat, st);
}

// ---------------------------------------------------------------------------
// chunk table: chunk c -> first object whose end lies beyond unit 64c
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_chunk_table_kernel(const u32x4* __restrict__ descs,
                                                                uint32_t n_obj,
                                                                uint32_t* __restrict__ chunk_first,
                                                                uint32_t n_chunks)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t g = c * kUnitsPerChunk;
    uint32_t lo = 0, hi = n_obj;                  // first o in [lo,hi] with end(o) > g
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        uint32_t s, e, lf;
        span_of(descs, mid, s, e, lf);
        if (e > g) hi = mid; else lo = mid + 1;
    }
    chunk_first[c] = lo;
}

// ---------------------------------------------------------------------------
// tile scheduling: which tiles a wavefront takes
// ---------------------------------------------------------------------------
struct TileIter {
    uint32_t tile, end, stride;
};

// variant 0: grid-stride over tiles (all wavefronts sweep the buffer together).
// variant 1: workgroups that share an XCD (equal blockIdx % 8) take one
//            contiguous eighth of the tiles, so the descriptor lines an XCD's
//            L2 holds are the ones its own wavefronts need next.
__device__ __forceinline__ TileIter make_tile_iter(uint32_t tile_begin, uint32_t tile_end, int variant)
{
    const uint32_t wave_in_block = threadIdx.x >> 6;
    TileIter it;
    if (variant == 1 && (gridDim.x & 7u) == 0) {
        const uint32_t xcd = blockIdx.x & 7u;
        const uint32_t blk = blockIdx.x >> 3;
        const uint32_t waves_per_xcd = (gridDim.x >> 3) * (kBlockThreads / 64);
        const uint32_t n = tile_end - tile_begin;
        const uint32_t per = (n + 7u) >> 3;
        uint32_t b = tile_begin + xcd * per;
        uint32_t e = b + per;
        if (b > tile_end) b = tile_end;
        if (e > tile_end) e = tile_end;
        it.tile = b + blk * (kBlockThreads / 64) + wave_in_block;
        it.end = e;
        it.stride = waves_per_xcd;
    } else {
        it.tile = tile_begin + blockIdx.x * (kBlockThreads / 64) + wave_in_block;
        it.end = tile_end;
        it.stride = gridDim.x * (kBlockThreads / 64);
    }
    return it;
}

// ---------------------------------------------------------------------------
// fill: Generate / generateDataUnit (pattern.go:73-116) for a whole batch
// ---------------------------------------------------------------------------
template <int UNROLL>
__global__ __launch_bounds__(256) void fill_kernel(const u32x4* __restrict__ descs,
                                                   const uint32_t* __restrict__ chunk_first,
                                                   uint32_t n_obj, uint32_t n_chunks,
                                                   u32x4* __restrict__ body,
                                                   uint32_t tile_begin, uint32_t tile_end, int variant)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};

    for (TileIter it = make_tile_iter(tile_begin, tile_end, variant); it.tile < it.end; it.tile += it.stride) {
        const uint32_t c = it.tile * kChunksPerTile + grp;
        const bool chunk_ok = c < n_chunks;
        const uint32_t g0 = c * kUnitsPerChunk;
        ObjState st;
        enter_object(descs, n_obj, chunk_ok ? chunk_first[c] : n_obj, j, cpat, st);

        for (uint32_t kb = 0; kb < kUnitsPerChunk; kb += UNROLL) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
                const uint32_t g = g0 + kb + i;
                if (__any(g >= st.end)) {
                    if (g >= st.end) advance_object(descs, n_obj, g, j, cpat, st);
                }
                if (g >= st.start) {              // inside the object (g < st.end holds here)
                    u32x4 e = st.tmpl;
                    e.x += g * gmul;              // lane 2: uint32(unitCount * 256), pattern.go:96,102
                    __builtin_nontemporal_store(e, &body[(size_t)g * 16u + (uint32_t)j]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// validate: Valid / validDataUnit (pattern.go:118-180) for a whole batch
// ---------------------------------------------------------------------------

// Per-group running outcome inside the current object.
struct RunState {
    uint32_t first;   // lowest global unit index with a reference-visible mismatch
    uint32_t mask;    // its OVAL_BAD_* bits
    uint32_t cnt;     // number of such units
    uint32_t sfirst;  // lowest global unit index with any mismatch (filler included)
};

__device__ __forceinline__ void reset(RunState& r) { r.first = kNone; r.mask = 0; r.cnt = 0; r.sfirst = kNone; }

// Lane 2 of a group publishes the group's outcome for object st.o.
__device__ __forceinline__ void flush_object(oval_obj_result* __restrict__ results,
                                             unsigned long long* __restrict__ counters,
                                             const ObjState& st, const RunState& r, bool leader)
{
    if (leader && (r.cnt != 0u || r.sfirst != kNone)) {
        oval_obj_result* R = results + st.o;
        if (r.cnt != 0u) {
            const unsigned long long key =
                ((unsigned long long)(r.first + st.ubias) << 32) | (unsigned long long)r.mask;
            atomicMin(reinterpret_cast<unsigned long long*>(R), key);
            const uint32_t old = atomicAdd(&R->bad_units, r.cnt);
            if (old == 0u && counters) atomicAdd(&counters[OVAL_CNT_OBJECTS_BAD], 1ull);
        }
        atomicMin(&R->strict_first_bad_unit, r.sfirst + st.ubias);
    }
}

// Exact per-field outcome of one lane's 16 bytes (only reached for a unit that
// failed the 128-bit compare or whose descriptor asks for the exact check).
__device__ uint32_t lane_field_mask(const u32x4& v, uint32_t ex, const ObjState& st, int j)
{
    uint32_t fm = 0;
    if (j == 0) {
        const int blen = (int)(st.lf & 0xFFu);
        const u32x4 t = st.tmpl;
        if (!trimmed_equals([&](int i) { return reg_byte(v, i); }, 16,
                            [&](int i) { return reg_byte(t, i); }, blen))
            fm |= OVAL_BAD_BUCKET;
    } else if (j == 1) {
        const int klen = (int)((st.lf >> 8) & 0xFFu);
        const u32x4 t = st.tmpl;
        if (!trimmed_equals([&](int i) { return reg_byte(v, i); }, 12,
                            [&](int i) { return reg_byte(t, i); }, klen))
            fm |= OVAL_BAD_KEY;
        if (v.w != st.tmpl.w) fm |= OVAL_BAD_WRITECNT;
    } else if (j == 2) {
        if (v.x != ex) fm |= OVAL_BAD_OFFSET;
        if (v.y != st.tmpl.y || st.force) fm |= OVAL_BAD_WORKER;
    } else {
        fm = OVAL_BAD_FILLER;
    }
    return fm;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void validate_kernel(const u32x4* __restrict__ descs,
                                                       const uint32_t* __restrict__ chunk_first,
                                                       uint32_t n_obj, uint32_t n_chunks,
                                                       uint32_t limit_units,
                                                       const u32x4* __restrict__ body,
                                                       uint32_t tile_begin, uint32_t tile_end,
                                                       oval_obj_result* __restrict__ results,
                                                       unsigned long long* __restrict__ counters,
                                                       int variant)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const uint32_t zwmask = (j == 2) ? 0u : 0xFFFFFFFFu;   // pattern.go:172 "Skip the unix time area."
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};

    // batch totals of this lane's group (meaningful in every lane of the group)
    uint32_t c_units = 0, c_bucket = 0, c_key = 0, c_wc = 0, c_off = 0, c_wid = 0, c_strict = 0;

    for (TileIter it = make_tile_iter(tile_begin, tile_end, variant); it.tile < it.end; it.tile += it.stride) {
        const uint32_t c = it.tile * kChunksPerTile + grp;
        const bool chunk_ok = c < n_chunks;
        const uint32_t g0 = c * kUnitsPerChunk;
        ObjState st;
        enter_object(descs, n_obj, chunk_ok ? chunk_first[c] : n_obj, j, cpat, st);
        RunState run;
        reset(run);

        for (uint32_t kb = 0; kb < kUnitsPerChunk; kb += UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
                const uint32_t g = g0 + kb + i;
                v[i] = (u32x4){0u, 0u, 0u, 0u};
                if (chunk_ok && g < limit_units)
                    v[i] = __builtin_nontemporal_load(&body[(size_t)g * 16u + (uint32_t)j]);
            }
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
                const uint32_t g = g0 + kb + i;
                if (__any(g >= st.end)) {
                    if (g >= st.end) {
                        flush_object(results, counters, st, run, j == 2);
                        reset(run);
                        advance_object(descs, n_obj, g, j, cpat, st);
                    }
                }
                const uint32_t ex = st.tmpl.x + g * gmul;
                const uint32_t d = (v[i].x ^ ex) | (v[i].y ^ st.tmpl.y) |
                                   (((v[i].z ^ st.tmpl.z) | (v[i].w ^ st.tmpl.w)) & zwmask);
                const bool bad = (g >= st.start) && ((d | st.force) != 0u);
                if (__any(bad)) {
                    uint32_t fm = bad ? lane_field_mask(v[i], ex, st, j) : 0u;
                    fm |= __shfl_xor(fm, 1);
                    fm |= __shfl_xor(fm, 2);
                    fm |= __shfl_xor(fm, 4);
                    fm |= __shfl_xor(fm, 8);      // now the unit's mask, in all 16 lanes of the group
                    const uint32_t ref = fm & OVAL_REF_FIELDS;
                    if (ref) {
                        if (run.cnt == 0u) { run.first = g; run.mask = ref; }
                        run.cnt++;
                        c_units++;
                        c_bucket += (ref & OVAL_BAD_BUCKET) ? 1u : 0u;
                        c_key += (ref & OVAL_BAD_KEY) ? 1u : 0u;
                        c_wc += (ref & OVAL_BAD_WRITECNT) ? 1u : 0u;
                        c_off += (ref & OVAL_BAD_OFFSET) ? 1u : 0u;
                        c_wid += (ref & OVAL_BAD_WORKER) ? 1u : 0u;
                    }
                    if (fm) {
                        if (run.sfirst == kNone) run.sfirst = g;
                        c_strict++;
                    }
                }
            }
        }

        // End of the tile.  When all four groups finished inside the same
        // object (large objects), min-reduce across the wavefront so that one
        // lane publishes; otherwise each group publishes its own.
        const uint32_t o_first = __builtin_amdgcn_readfirstlane(st.o);
        if (__all(st.o == o_first)) {
            if (o_first < n_obj && __any(run.cnt != 0u || run.sfirst != kNone)) {
                RunState w = run;
                // ubias is object-wide, take it from lane 2 of group 0
                ObjState ws = st;
                ws.ubias = __shfl(st.ubias, 2);
                uint32_t f = w.first;
                f = min(f, (uint32_t)__shfl_xor(f, 16));
                f = min(f, (uint32_t)__shfl_xor(f, 32));
                uint32_t m = (w.first == f) ? w.mask : 0u;   // a unit belongs to exactly one group
                m |= __shfl_xor(m, 16);
                m |= __shfl_xor(m, 32);
                uint32_t n = w.cnt;
                n += __shfl_xor(n, 16);
                n += __shfl_xor(n, 32);
                uint32_t sf = w.sfirst;
                sf = min(sf, (uint32_t)__shfl_xor(sf, 16));
                sf = min(sf, (uint32_t)__shfl_xor(sf, 32));
                w.first = f; w.mask = m; w.cnt = n; w.sfirst = sf;
                flush_object(results, counters, ws, w, lane == 2);
            }
        } else {
            flush_object(results, counters, st, run, j == 2);
        }
    }

    if (counters && __any(c_strict != 0u)) {
        // sum the four groups, then lane 2 adds the wavefront's totals
        uint32_t t[7] = {c_units, c_bucket, c_key, c_wc, c_off, c_wid, c_strict};
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            t[k] += __shfl_xor(t[k], 16);
            t[k] += __shfl_xor(t[k], 32);
        }
        if (lane == 2) {
            if (t[0]) atomicAdd(&counters[OVAL_CNT_UNITS_BAD], (unsigned long long)t[0]);
            if (t[1]) atomicAdd(&counters[OVAL_CNT_BAD_BUCKET], (unsigned long long)t[1]);
            if (t[2]) atomicAdd(&counters[OVAL_CNT_BAD_KEY], (unsigned long long)t[2]);
            if (t[3]) atomicAdd(&counters[OVAL_CNT_BAD_WRITECNT], (unsigned long long)t[3]);
            if (t[4]) atomicAdd(&counters[OVAL_CNT_BAD_OFFSET], (unsigned long long)t[4]);
            if (t[5]) atomicAdd(&counters[OVAL_CNT_BAD_WORKER], (unsigned long long)t[5]);
            if (t[6]) atomicAdd(&counters[OVAL_CNT_STRICT_UNITS_BAD], (unsigned long long)t[6]);
        }
    }
}

// ===========================================================================
// Wave-walk kernels.
//
// One wavefront instruction covers four CONSECUTIVE units (1 KiB contiguous),
// a wavefront pass covers one 64 KiB tile, and the walk over objects is done
// once per wavefront on the scalar unit: descriptors are fetched with scalar
// loads (constant address space), so waiting for one never drains the vector
// memory queue that holds the body's stores or prefetched loads.
// ===========================================================================
typedef const __attribute__((address_space(4))) u32x4* sdesc_ptr;

// Wave-uniform view of the current object.
struct WObj {
    uint32_t o, start, end, lf, ubias;
    u32x4 d0, d1, d2;          // bytes [0,48) of the descriptor = header template
};

// A descriptor fetched ahead of need (one s_load_dwordx16), not yet decoded.
struct WRaw {
    uint32_t o;
    u32x4 d0, d1, d2, d3;
};

// n_obj >= 1.  The load is unconditional (index clamped to the last object) so
// that nothing touches the destination registers before the values are needed:
// the wait for it then sits at the next object switch, not here.
__device__ __forceinline__ void w_fetch(sdesc_ptr descs, uint32_t n_obj, uint32_t o, WRaw& r)
{
    const uint32_t oc = o < n_obj ? o : n_obj - 1u;
    r.o = o < n_obj ? o : n_obj;
    r.d0 = descs[4u * oc];
    r.d1 = descs[4u * oc + 1u];
    r.d2 = descs[4u * oc + 2u];
    r.d3 = descs[4u * oc + 3u];
}

__device__ __forceinline__ void w_activate(const WRaw& r, uint32_t n_obj, WObj& w)
{
    // Every fetched register counts as read here, so none of them is handed
    // out as a temporary (and waited for) while the fetch is still in flight.
    asm volatile("" ::"s"(r.d0), "s"(r.d1), "s"(r.d2), "s"(r.d3));
    w.o = r.o;
    w.d0 = r.d0; w.d1 = r.d1; w.d2 = r.d2;
    if (r.o < n_obj) {
        w.start = (r.d3.x >> 8) | (r.d3.y << 24);              // body_offset / 256
        w.end = w.start + r.d3.z;
        w.lf = r.d3.w;
        w.ubias = r.d2.x - w.start;                            // first_unit - start
    } else {
        w.start = kNone; w.end = kNone; w.lf = 0; w.ubias = 0;
    }
}

// Make object `o` current and start fetching its successor.
__device__ __forceinline__ void w_enter(sdesc_ptr descs, uint32_t n_obj, uint32_t o, WObj& w, WRaw& next)
{
    WRaw r;
    w_fetch(descs, n_obj, o, r);
    w_fetch(descs, n_obj, o + 1u, next);
    w_activate(r, n_obj, w);
}

// Stepping to the successor is done in two halves by the kernels: w_activate
// + rebuilding the lane template (which reads the fetched registers), then,
// behind a scheduling barrier, w_fetch of the one after it.  Issued earlier,
// the fetch would need registers of its own and be copied (and waited for)
// on the spot.
// This lane's expected 16 bytes for unit index 0 of the walk (lane 2's .x gets
// + (g << 8) per unit).
__device__ __forceinline__ u32x4 w_lane_template(const WObj& w, int j, const u32x4& cpat)
{
    u32x4 t = cpat;
    if (j == 0) t = w.d0;
    if (j == 1) t = w.d1;
    if (j == 2) { t = w.d2; t.x = w.ubias << 8; }
    return t;
}

__device__ __forceinline__ uint32_t w_lane_force(const WObj& w, int j)
{
    return (j < 3) ? ((w.lf >> 16) & (1u << j)) : 0u;
}

__device__ __forceinline__ uint32_t wave_tile_begin(uint32_t tile_begin, uint32_t tile_end, int variant,
                                                    uint32_t& end, uint32_t& stride)
{
    TileIter it = make_tile_iter(tile_begin, tile_end, variant);
    end = __builtin_amdgcn_readfirstlane(it.end);
    stride = __builtin_amdgcn_readfirstlane(it.stride);
    return __builtin_amdgcn_readfirstlane(it.tile);
}

template <int UNROLL>
__global__ __launch_bounds__(256) void fill_wave_kernel(sdesc_ptr descs, const uint32_t* __restrict__ tile_first,
                                                        uint32_t n_obj, u32x4* __restrict__ body,
                                                        uint32_t tile_begin, uint32_t tile_end, int variant)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};
    uint32_t t_end, t_stride;
    for (uint32_t tile = wave_tile_begin(tile_begin, tile_end, variant, t_end, t_stride); tile < t_end;
         tile += t_stride) {
        const uint32_t g_tile = tile * (kUnitsPerChunk * kChunksPerTile);
        WObj w;
        WRaw next;
        w_enter(descs, n_obj, __builtin_amdgcn_readfirstlane(tile_first[tile * kChunksPerTile]), w, next);
        u32x4 tmpl = w_lane_template(w, j, cpat);
        u32x4* const tile_ptr = body + (size_t)g_tile * 16u + (uint32_t)lane;
#pragma unroll UNROLL
        for (uint32_t k = 0; k < kUnitsPerChunk; ++k) {
            const uint32_t g4 = g_tile + 4u * k;
            const uint32_t g = g4 + grp;
            if (g4 >= w.start && g4 + 4u <= w.end) {            // all four units inside the object
                u32x4 e = tmpl;
                e.x += g * gmul;
                tile_ptr[k * 64u] = e;
            } else {
                for (;;) {
                    if (g >= w.start && g < w.end) {
                        u32x4 e = tmpl;
                        e.x += g * gmul;
                        tile_ptr[k * 64u] = e;
                    }
                    if (w.end >= g4 + 4u) break;                 // kNone for "no more objects"
                    w_activate(next, n_obj, w);
                    tmpl = w_lane_template(w, j, cpat);
                    __builtin_amdgcn_sched_barrier(0);
                    w_fetch(descs, n_obj, w.o + 1u, next);
                }
            }
        }
    }
}

// Running outcome for the current object, wave-uniform.
struct WRun {
    uint32_t first, mask, cnt, sfirst;
};

struct WTotals {
    uint32_t units, bucket, key, wc, off, wid, strict;
};

__device__ __forceinline__ void w_flush(oval_obj_result* __restrict__ results,
                                        unsigned long long* __restrict__ counters, const WObj& w, WRun& r,
                                        int lane)
{
    if (r.cnt != 0u || r.sfirst != kNone) {
        if (lane == 0) {
            oval_obj_result* R = results + w.o;
            if (r.cnt != 0u) {
                const unsigned long long key =
                    ((unsigned long long)(r.first + w.ubias) << 32) | (unsigned long long)r.mask;
                atomicMin(reinterpret_cast<unsigned long long*>(R), key);
                const uint32_t old = atomicAdd(&R->bad_units, r.cnt);
                if (old == 0u && counters) atomicAdd(&counters[OVAL_CNT_OBJECTS_BAD], 1ull);
            }
            atomicMin(&R->strict_first_bad_unit, r.sfirst + w.ubias);
        }
        r.first = kNone; r.mask = 0; r.cnt = 0; r.sfirst = kNone;
    }
}

// Exact verdict for the lanes flagged `bad` in one wavefront instruction (four
// units, all inside object w): per-field masks, then a ballot over the
// wavefront gives the lowest bad unit and the number of bad units.
__device__ __forceinline__ void w_handle_bad(const u32x4& v, uint32_t ex, const u32x4& tmpl, uint32_t lf,
                                          uint32_t force, bool bad, int j, uint32_t g4, WRun& r, WTotals& t)
{
    uint32_t fm = 0;
    if (bad) {
        ObjState st;
        st.tmpl = tmpl; st.lf = lf; st.force = force;
        st.o = 0; st.start = 0; st.end = 0; st.ubias = 0;
        fm = lane_field_mask(v, ex, st, j);
    }
    fm |= __shfl_xor(fm, 1);
    fm |= __shfl_xor(fm, 2);
    fm |= __shfl_xor(fm, 4);
    fm |= __shfl_xor(fm, 8);                                    // the unit's mask in all 16 lanes of its group
    const uint32_t ref = fm & OVAL_REF_FIELDS;
    const unsigned long long refb = __ballot(ref != 0u);
    const unsigned long long anyb = __ballot(fm != 0u);
    if (refb) {
        const uint32_t first_lane = (uint32_t)__builtin_ctzll(refb);
        if (r.cnt == 0u) {
            r.first = g4 + (first_lane >> 4);
            r.mask = (uint32_t)__builtin_amdgcn_readlane((int)ref, (int)first_lane);
        }
        const uint32_t n = (uint32_t)__builtin_popcountll(refb) >> 4;
        r.cnt += n;
        t.units += n;
        t.bucket += (uint32_t)__builtin_popcountll(__ballot((ref & OVAL_BAD_BUCKET) != 0u)) >> 4;
        t.key += (uint32_t)__builtin_popcountll(__ballot((ref & OVAL_BAD_KEY) != 0u)) >> 4;
        t.wc += (uint32_t)__builtin_popcountll(__ballot((ref & OVAL_BAD_WRITECNT) != 0u)) >> 4;
        t.off += (uint32_t)__builtin_popcountll(__ballot((ref & OVAL_BAD_OFFSET) != 0u)) >> 4;
        t.wid += (uint32_t)__builtin_popcountll(__ballot((ref & OVAL_BAD_WORKER) != 0u)) >> 4;
    }
    if (anyb) {
        if (r.sfirst == kNone) r.sfirst = g4 + ((uint32_t)__builtin_ctzll(anyb) >> 4);
        t.strict += (uint32_t)__builtin_popcountll(anyb) >> 4;
    }
}

__device__ __forceinline__ uint32_t w_diff(const u32x4& v, uint32_t ex, const u32x4& tmpl, uint32_t zwmask)
{
    return (v.x ^ ex) | (v.y ^ tmpl.y) | (((v.z ^ tmpl.z) | (v.w ^ tmpl.w)) & zwmask);
}

template <int UNROLL>
__global__ __launch_bounds__(256) void validate_wave_kernel(sdesc_ptr descs,
                                                            const uint32_t* __restrict__ tile_first,
                                                            uint32_t n_obj, uint32_t limit_units,
                                                            const u32x4* __restrict__ body,
                                                            uint32_t tile_begin, uint32_t tile_end,
                                                            oval_obj_result* __restrict__ results,
                                                            unsigned long long* __restrict__ counters,
                                                            int variant)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const uint32_t zwmask = (j == 2) ? 0u : 0xFFFFFFFFu;       // pattern.go:172 "Skip the unix time area."
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};
    WTotals tot = {0, 0, 0, 0, 0, 0, 0};
    uint32_t t_end, t_stride;
    for (uint32_t tile = wave_tile_begin(tile_begin, tile_end, variant, t_end, t_stride); tile < t_end;
         tile += t_stride) {
        const uint32_t g_tile = tile * (kUnitsPerChunk * kChunksPerTile);
        WObj w;
        WRaw next;
        w_enter(descs, n_obj, __builtin_amdgcn_readfirstlane(tile_first[tile * kChunksPerTile]), w, next);
        u32x4 tmpl = w_lane_template(w, j, cpat);
        uint32_t force = w_lane_force(w, j);
        WRun run = {kNone, 0, 0, kNone};
        const u32x4* const tile_ptr = body + (size_t)g_tile * 16u + (uint32_t)lane;

        for (uint32_t kb = 0; kb < kUnitsPerChunk; kb += UNROLL) {
            bool careful = (run.cnt != 0u) || (run.sfirst != kNone);
            const uint32_t o_batch = w.o;
            if (!careful) {
                // pass 1: stream the batch; no side effects until it is known to be clean
                u32x4 v[UNROLL];
#pragma unroll
                for (int i = 0; i < UNROLL; ++i) {
                    v[i] = (u32x4){0u, 0u, 0u, 0u};
                    if (g_tile + 4u * (kb + i) + grp < limit_units)
                        v[i] = __builtin_nontemporal_load(tile_ptr + (kb + i) * 64u);
                }
                uint32_t anybad = 0;
#pragma unroll
                for (int i = 0; i < UNROLL; ++i) {
                    const uint32_t g4 = g_tile + 4u * (kb + i);
                    const uint32_t g = g4 + grp;
                    if (g4 >= w.start && g4 + 4u <= w.end) {
                        anybad |= w_diff(v[i], tmpl.x + g * gmul, tmpl, zwmask) | force;
                    } else {
                        for (;;) {
                            if (g >= w.start && g < w.end)
                                anybad |= w_diff(v[i], tmpl.x + g * gmul, tmpl, zwmask) | force;
                            if (w.end >= g4 + 4u) break;
                            w_activate(next, n_obj, w);
                            tmpl = w_lane_template(w, j, cpat);
                            force = w_lane_force(w, j);
                            __builtin_amdgcn_sched_barrier(0);
                            w_fetch(descs, n_obj, w.o + 1u, next);
                        }
                    }
                }
                if (__any(anybad != 0u)) {
                    careful = true;                             // rewind to the batch's first object
                    w_enter(descs, n_obj, o_batch, w, next);
                    tmpl = w_lane_template(w, j, cpat);
                    force = w_lane_force(w, j);
                }
            }
            if (careful) {
                // pass 2: the same units one instruction at a time, with the exact verdicts
#pragma unroll 1
                for (int i = 0; i < UNROLL; ++i) {
                    const uint32_t g4 = g_tile + 4u * (kb + i);
                    const uint32_t g = g4 + grp;
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (g < limit_units) v = __builtin_nontemporal_load(tile_ptr + (kb + i) * 64u);
                    for (;;) {
                        const uint32_t ex = tmpl.x + g * gmul;
                        const bool bad = (g >= w.start && g < w.end) &&
                                         ((w_diff(v, ex, tmpl, zwmask) | force) != 0u);
                        if (__any(bad)) w_handle_bad(v, ex, tmpl, w.lf, force, bad, j, g4, run, tot);
                        if (w.end >= g4 + 4u) break;
                        w_flush(results, counters, w, run, lane);
                        w_activate(next, n_obj, w);
                        tmpl = w_lane_template(w, j, cpat);
                        force = w_lane_force(w, j);
                        __builtin_amdgcn_sched_barrier(0);
                        w_fetch(descs, n_obj, w.o + 1u, next);
                    }
                }
            }
        }
        if (w.o < n_obj) w_flush(results, counters, w, run, lane);
    }
    if (counters && tot.strict != 0u && lane == 0) {
        if (tot.units) atomicAdd(&counters[OVAL_CNT_UNITS_BAD], (unsigned long long)tot.units);
        if (tot.bucket) atomicAdd(&counters[OVAL_CNT_BAD_BUCKET], (unsigned long long)tot.bucket);
        if (tot.key) atomicAdd(&counters[OVAL_CNT_BAD_KEY], (unsigned long long)tot.key);
        if (tot.wc) atomicAdd(&counters[OVAL_CNT_BAD_WRITECNT], (unsigned long long)tot.wc);
        if (tot.off) atomicAdd(&counters[OVAL_CNT_BAD_OFFSET], (unsigned long long)tot.off);
        if (tot.wid) atomicAdd(&counters[OVAL_CNT_BAD_WORKER], (unsigned long long)tot.wid);
        atomicAdd(&counters[OVAL_CNT_STRICT_UNITS_BAD], (unsigned long long)tot.strict);
    }
}

// results[i] = {none, none, 0, none}; counters += objects/units checked
__global__ __launch_bounds__(256) void init_results_kernel(u32x4* __restrict__ results, uint32_t n_obj,
                                                           unsigned long long* __restrict__ counters,
                                                           unsigned long long total_units)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_obj; i += stride)
        results[i] = (u32x4){kNone, kNone, 0u, kNone};
    if (counters && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&counters[OVAL_CNT_OBJECTS_CHECKED], (unsigned long long)n_obj);
        atomicAdd(&counters[OVAL_CNT_UNITS_CHECKED], total_units);
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_build_chunk_table(hipStream_t s, const void* descs, uint32_t n_obj,
                                    uint32_t* chunk_first, uint32_t n_chunks)
{
    if (n_chunks == 0) return hipSuccess;
    const uint32_t blocks = (n_chunks + 255u) / 256u;
    build_chunk_table_kernel<<<blocks, 256, 0, s>>>(static_cast<const u32x4*>(descs), n_obj,
                                                    chunk_first, n_chunks);
    return hipGetLastError();
}

static int clamp_grid(int want, uint32_t n_tiles)
{
    const uint32_t need = (n_tiles + (kBlockThreads / 64) - 1) / (kBlockThreads / 64);
    if (need < (uint32_t)want) {
        // keep the grid a multiple of 8 so the XCD mapping stays balanced
        uint32_t g = (need + 7u) & ~7u;
        return (int)(g ? g : 8u);
    }
    return want;
}

hipError_t launch_fill(hipStream_t s, const LaunchCfg& cfg, const LayoutView& lv, void* body,
                       uint32_t tile_begin, uint32_t tile_end)
{
    if (tile_end <= tile_begin) return hipSuccess;
    const int grid = clamp_grid(cfg.grid_blocks, tile_end - tile_begin);
    const int variant = cfg.variant & 1;
    if (!(cfg.variant & 4)) {
        sdesc_ptr sd = (sdesc_ptr)lv.descs;
        u32x4* b = static_cast<u32x4*>(body);
#define OVAL_LAUNCH_FILL(U) fill_wave_kernel<U><<<grid, kBlockThreads, 0, s>>>(sd, lv.chunk_first, lv.n_obj, b, tile_begin, tile_end, variant)
        switch ((cfg.variant >> 4) & 3) {
        case 1: OVAL_LAUNCH_FILL(2); break;
        case 2: OVAL_LAUNCH_FILL(8); break;
        case 3: OVAL_LAUNCH_FILL(16); break;
        default: OVAL_LAUNCH_FILL(4); break;
        }
#undef OVAL_LAUNCH_FILL
        return hipGetLastError();
    }
    if (cfg.variant & 2)
        fill_kernel<8><<<grid, kBlockThreads, 0, s>>>(static_cast<const u32x4*>(lv.descs), lv.chunk_first,
                                                      lv.n_obj, lv.n_chunks, static_cast<u32x4*>(body),
                                                      tile_begin, tile_end, variant);
    else
        fill_kernel<4><<<grid, kBlockThreads, 0, s>>>(static_cast<const u32x4*>(lv.descs), lv.chunk_first,
                                                      lv.n_obj, lv.n_chunks, static_cast<u32x4*>(body),
                                                      tile_begin, tile_end, variant);
    return hipGetLastError();
}

hipError_t launch_init_results(hipStream_t s, const LayoutView& lv, oval_obj_result* results,
                               unsigned long long* counters)
{
    uint32_t blocks = (lv.n_obj + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0u) blocks = 1u;
    init_results_kernel<<<blocks, 256, 0, s>>>(reinterpret_cast<u32x4*>(results), lv.n_obj, counters,
                                               (unsigned long long)lv.total_units);
    return hipGetLastError();
}

hipError_t launch_validate(hipStream_t s, const LaunchCfg& cfg, const LayoutView& lv,
                           const void* body, uint32_t tile_begin, uint32_t tile_end,
                           oval_obj_result* results, unsigned long long* counters)
{
    if (tile_end <= tile_begin) return hipSuccess;
    const int grid = clamp_grid(cfg.grid_blocks, tile_end - tile_begin);
    const int variant = cfg.variant & 1;
    if (!(cfg.variant & 4)) {
        sdesc_ptr sd = (sdesc_ptr)lv.descs;
        const u32x4* b = static_cast<const u32x4*>(body);
#define OVAL_LAUNCH_VAL(U) validate_wave_kernel<U><<<grid, kBlockThreads, 0, s>>>(sd, lv.chunk_first, lv.n_obj, lv.span_units, b, tile_begin, tile_end, results, counters, variant)
        switch ((cfg.variant >> 4) & 3) {
        case 1: OVAL_LAUNCH_VAL(2); break;
        case 2: OVAL_LAUNCH_VAL(4); break;
        case 3: OVAL_LAUNCH_VAL(16); break;
        default: OVAL_LAUNCH_VAL(8); break;
        }
#undef OVAL_LAUNCH_VAL
        return hipGetLastError();
    }
    if (cfg.variant & 2)
        validate_kernel<8><<<grid, kBlockThreads, 0, s>>>(
            static_cast<const u32x4*>(lv.descs), lv.chunk_first, lv.n_obj, lv.n_chunks, lv.span_units,
            static_cast<const u32x4*>(body), tile_begin, tile_end, results, counters, variant);
    else
        validate_kernel<4><<<grid, kBlockThreads, 0, s>>>(
            static_cast<const u32x4*>(lv.descs), lv.chunk_first, lv.n_obj, lv.n_chunks, lv.span_units,
            static_cast<const u32x4*>(body), tile_begin, tile_end, results, counters, variant);
    return hipGetLastError();
}

}  // namespace oval
