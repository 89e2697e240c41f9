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
