// oval_kernels.hip — gfx950 (CDNA4, wave64) kernels of the data-pattern engine.
//
// A 256-byte data unit (internal/pattern/pattern.go:94-116) is 16 lanes x 16
// bytes, so one wavefront instruction moves four CONSECUTIVE units (1 KiB
// contiguous) with 128-bit accesses.  Lane j of a 16-lane group owns bytes
// [16j, 16j+16) of its unit:
//   lane 0      bucket name field                    (pattern.go:98)
//   lane 1      key field (12 B) + write count       (pattern.go:98,101)
//   lane 2      offset in object, worker ID, time    (pattern.go:102-106)
//   lanes 3-15  filler, byte value == byte offset    (pattern.go:109-114)
// Lanes 0..2 keep their slice of the object's 48-byte header template in
// registers, lanes 3..15 keep constants computed from the lane index; nothing
// but the body itself is streamed from or to HBM.
//
// A wavefront pass covers one or more 64 KiB tiles.  The walk over objects is
// done once per wavefront on the scalar unit: descriptors are fetched with
// scalar loads (constant address space, one s_load_dwordx16 each) and the next
// one is prefetched into SGPRs, so waiting for a descriptor never drains the
// vector-memory queue that holds the body's stores or prefetched loads.
//
// Stores are non-temporal: a fill that leaves its lines dirty in L2 / Infinity
// Cache makes the validate pass that follows pay for their write-back (measured:
// validate 6.9 -> 5.3 TB/s behind a plain-store fill).
#include "oval_kernels.h"
#include "oval_trimspace.h"

namespace oval {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4* sdesc_ptr;

constexpr uint32_t kNone = 0xFFFFFFFFu;

// bytes 4k..4k+3 of the filler, as a little-endian dword
__device__ __forceinline__ uint32_t filler_dword(uint32_t k) { return 0x03020100u + 0x04040404u * k; }

__device__ __forceinline__ uint32_t reg_byte(const u32x4& v, int i)
{
    uint32_t w = (i & 8) ? ((i & 4) ? v.w : v.z) : ((i & 4) ? v.y : v.x);
    return (w >> ((i & 3) * 8)) & 0xFFu;
}

// ---------------------------------------------------------------------------
// tile table: tile t -> first object whose end lies beyond unit 256 t
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_tile_table_kernel(const u32x4* __restrict__ descs,
                                                               uint32_t n_obj,
                                                               uint32_t* __restrict__ tile_first,
                                                               uint32_t n_tiles)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t g = t * kUnitsPerTile;
    uint32_t lo = 0, hi = n_obj;                  // first o in [lo,hi] with end(o) > g
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const u32x4 s3 = descs[4u * mid + 3u];    // body_offset (8) | n_units (4) | lens+flags (4)
        const uint32_t end = ((s3.x >> 8) | (s3.y << 24)) + s3.z;
        if (end > g) hi = mid; else lo = mid + 1;
    }
    tile_first[t] = lo;
}

// ---------------------------------------------------------------------------
// which passes a wavefront takes
// ---------------------------------------------------------------------------
// variant 0: grid-stride (all wavefronts sweep the buffer together).
// variant 1: workgroups that share an XCD (equal blockIdx % 8) take one
//            contiguous eighth of the passes.
__device__ __forceinline__ uint32_t first_pass(uint32_t begin, uint32_t end_in, int variant,
                                               uint32_t& end, uint32_t& stride)
{
    const uint32_t wave_in_block = threadIdx.x >> 6;
    const uint32_t wpb = kBlockThreads / 64;
    uint32_t first;
    if (variant == 1 && (gridDim.x & 7u) == 0) {
        const uint32_t xcd = blockIdx.x & 7u;
        const uint32_t blk = blockIdx.x >> 3;
        const uint32_t per = (end_in - begin + 7u) >> 3;
        uint32_t b = begin + xcd * per;
        uint32_t e = b + per;
        if (b > end_in) b = end_in;
        if (e > end_in) e = end_in;
        first = b + blk * wpb + wave_in_block;
        end = e;
        stride = (gridDim.x >> 3) * wpb;
    } else {
        first = begin + blockIdx.x * wpb + wave_in_block;
        end = end_in;
        stride = gridDim.x * wpb;
    }
    // wave-uniform by construction; tell the compiler so (scalar loop control)
    end = __builtin_amdgcn_readfirstlane(end);
    stride = __builtin_amdgcn_readfirstlane(stride);
    return __builtin_amdgcn_readfirstlane(first);
}

// ---------------------------------------------------------------------------
// wave-uniform object walk
// ---------------------------------------------------------------------------
struct WObj {                    // the current object
    uint32_t o, start, end, lf, ubias;   // ubias = first_unit - start, so unitCount = g + ubias
    u32x4 d0, d1, d2;            // bytes [0,48) of the descriptor = header template
};

struct WRaw {                    // a descriptor fetched ahead of need, not yet decoded
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
        w.lf = r.d3.w;                                         // bucket_len | key_len << 8 | flags << 16
        w.ubias = r.d2.x - w.start;                            // first_unit - start
    } else {
        w.start = kNone; w.end = kNone; w.lf = 0; w.ubias = 0; // no more objects
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

// This lane's expected 16 bytes; lane 2's .x gets + (g << 8) per unit, which
// makes uint32(unitCount * 256) (pattern.go:96,102).
__device__ __forceinline__ u32x4 w_lane_template(const WObj& w, int j, const u32x4& cpat)
{
    u32x4 t = cpat;
    if (j == 0) t = w.d0;
    if (j == 1) t = w.d1;
    if (j == 2) { t = w.d2; t.x = w.ubias << 8; }
    return t;
}

// Non-zero when this lane's field needs the exact check on every unit:
// descriptor flags BUCKET_SLOW (lane 0), KEY_SLOW (lane 1), WORKER_OOR (lane 2).
__device__ __forceinline__ uint32_t w_lane_force(const WObj& w, int j)
{
    return (j < 3) ? ((w.lf >> 16) & (1u << j)) : 0u;
}

// ---------------------------------------------------------------------------
// fill: Generate / generateDataUnit (pattern.go:73-116) for a whole batch
// ---------------------------------------------------------------------------
template <int UNROLL, int PASS>   // PASS = tiles per wavefront pass
__global__ __launch_bounds__(256) void fill_kernel(sdesc_ptr descs, const uint32_t* __restrict__ tile_first,
                                                   uint32_t n_obj, u32x4* __restrict__ body,
                                                   uint32_t tile_begin, uint32_t tile_end, int variant)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};
    uint32_t p_end, p_stride;
    for (uint32_t pass = first_pass(tile_begin / PASS, (tile_end + PASS - 1) / PASS, variant, p_end, p_stride);
         pass < p_end; pass += p_stride) {
        const uint32_t tile = pass * PASS;
        const uint32_t g_tile = tile * kUnitsPerTile;
        WObj w;
        WRaw next;
        w_enter(descs, n_obj, __builtin_amdgcn_readfirstlane(tile_first[tile]), w, next);
        u32x4 tmpl = w_lane_template(w, j, cpat);
        u32x4* const tile_ptr = body + (size_t)g_tile * 16u + (uint32_t)lane;
#pragma unroll UNROLL
        for (uint32_t k = 0; k < (kUnitsPerTile / 4u) * PASS; ++k) {
            const uint32_t g4 = g_tile + 4u * k;
            const uint32_t g = g4 + grp;
            if (g4 >= w.start && g4 + 4u <= w.end) {            // all four units inside the object
                u32x4 e = tmpl;
                e.x += g * gmul;
                __builtin_nontemporal_store(e, tile_ptr + (size_t)k * 64u);
            } else {                                            // boundary, gap, or past the last object
                for (;;) {
                    if (g >= w.start && g < w.end) {
                        u32x4 e = tmpl;
                        e.x += g * gmul;
                        __builtin_nontemporal_store(e, tile_ptr + (size_t)k * 64u);
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

// ---------------------------------------------------------------------------
// validate: Valid / validDataUnit (pattern.go:118-180) for a whole batch
// ---------------------------------------------------------------------------

// Running outcome for the current object, wave-uniform.
struct WRun {
    uint32_t first;   // lowest global unit index with a reference-visible mismatch
    uint32_t mask;    // its OVAL_BAD_* bits
    uint32_t cnt;     // number of such units
    uint32_t sfirst;  // lowest global unit index with any mismatch (filler included)
};

struct WTotals {
    uint32_t units, bucket, key, wc, off, wid, strict;
};

// One lane publishes the wavefront's outcome for object w.o: a 64-bit
// atomicMin keeps the lowest bad unit together with its field mask.
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

// Exact per-field outcome of one lane's 16 bytes (pattern.go:140-170); only
// reached for a unit that failed the 128-bit compare or whose descriptor asks
// for the exact check.
__device__ __forceinline__ uint32_t lane_field_mask(const u32x4& v, uint32_t ex, const u32x4& tmpl, uint32_t lf,
                                                    uint32_t force, int j)
{
    uint32_t fm = 0;
    if (j == 0) {
        const int blen = (int)(lf & 0xFFu);
        if (!trimmed_equals([&](int i) { return reg_byte(v, i); }, 16,
                            [&](int i) { return reg_byte(tmpl, i); }, blen))
            fm |= OVAL_BAD_BUCKET;
    } else if (j == 1) {
        const int klen = (int)((lf >> 8) & 0xFFu);
        if (!trimmed_equals([&](int i) { return reg_byte(v, i); }, 12,
                            [&](int i) { return reg_byte(tmpl, i); }, klen))
            fm |= OVAL_BAD_KEY;
        if (v.w != tmpl.w) fm |= OVAL_BAD_WRITECNT;
    } else if (j == 2) {
        if (v.x != ex) fm |= OVAL_BAD_OFFSET;
        if (v.y != tmpl.y || force) fm |= OVAL_BAD_WORKER;     // force: worker ID outside uint32
    } else {
        fm = OVAL_BAD_FILLER;
    }
    return fm;
}

// Exact verdict for the lanes flagged `bad` in one wavefront instruction (four
// units, the flagged ones all inside object w): per-field masks, then a ballot
// over the wavefront gives the lowest bad unit and the number of bad units.
__device__ __forceinline__ void w_handle_bad(const u32x4& v, uint32_t ex, const u32x4& tmpl, uint32_t lf,
                                             uint32_t force, bool bad, int j, uint32_t g4, WRun& r, WTotals& t)
{
    uint32_t fm = bad ? lane_field_mask(v, ex, tmpl, lf, force, j) : 0u;
    fm |= __shfl_xor(fm, 1);
    fm |= __shfl_xor(fm, 2);
    fm |= __shfl_xor(fm, 4);
    fm |= __shfl_xor(fm, 8);                                    // the unit's mask in all 16 lanes of its group
    const uint32_t ref = fm & OVAL_REF_FIELDS;
    const unsigned long long refb = __ballot(ref != 0u);
    const unsigned long long anyb = __ballot(fm != 0u);
    if (refb) {
        const uint32_t first_lane = (uint32_t)__builtin_ctzll(refb);
        if (r.cnt == 0u) {                                      // units arrive in ascending order
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

template <int UNROLL, int PASS>
__global__ __launch_bounds__(256) void validate_kernel(sdesc_ptr descs, const uint32_t* __restrict__ tile_first,
                                                       uint32_t n_obj, uint32_t limit_units,
                                                       const u32x4* __restrict__ body,
                                                       uint32_t tile_begin, uint32_t tile_end,
                                                       oval_obj_result* __restrict__ results,
                                                       unsigned long long* __restrict__ counters, int variant)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const uint32_t zwmask = (j == 2) ? 0u : 0xFFFFFFFFu;       // pattern.go:172 "Skip the unix time area."
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};
    WTotals tot = {0, 0, 0, 0, 0, 0, 0};
    uint32_t p_end, p_stride;
    for (uint32_t pass = first_pass(tile_begin / PASS, (tile_end + PASS - 1) / PASS, variant, p_end, p_stride);
         pass < p_end; pass += p_stride) {
        const uint32_t tile = pass * PASS;
        const uint32_t g_tile = tile * kUnitsPerTile;
        WObj w;
        WRaw next;
        w_enter(descs, n_obj, __builtin_amdgcn_readfirstlane(tile_first[tile]), w, next);
        u32x4 tmpl = w_lane_template(w, j, cpat);
        uint32_t force = w_lane_force(w, j);
        WRun run = {kNone, 0, 0, kNone};
        const u32x4* const tile_ptr = body + (size_t)g_tile * 16u + (uint32_t)lane;

        for (uint32_t kb = 0; kb < (kUnitsPerTile / 4u) * PASS; kb += UNROLL) {
            bool careful = (run.cnt != 0u) || (run.sfirst != kNone);
            const uint32_t o_batch = w.o;
            if (!careful) {
                // pass 1: stream the batch; no side effects until it is known to be clean
                u32x4 v[UNROLL];
#pragma unroll
                for (int i = 0; i < UNROLL; ++i) {
                    v[i] = (u32x4){0u, 0u, 0u, 0u};
                    if (g_tile + 4u * (kb + i) + grp < limit_units)
                        v[i] = __builtin_nontemporal_load(tile_ptr + (size_t)(kb + i) * 64u);
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
                    if (g < limit_units) v = __builtin_nontemporal_load(tile_ptr + (size_t)(kb + i) * 64u);
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
hipError_t launch_build_tile_table(hipStream_t s, const void* descs, uint32_t n_obj,
                                   uint32_t* tile_first, uint32_t n_tiles)
{
    if (n_tiles == 0) return hipSuccess;
    const uint32_t blocks = (n_tiles + 255u) / 256u;
    build_tile_table_kernel<<<blocks, 256, 0, s>>>(static_cast<const u32x4*>(descs), n_obj, tile_first, n_tiles);
    return hipGetLastError();
}

static int clamp_grid(int want, uint32_t n_passes)
{
    const uint32_t wpb = kBlockThreads / 64;
    const uint32_t need = (n_passes + wpb - 1) / wpb;
    if (need < (uint32_t)want) {
        // keep the grid a multiple of 8 so the XCD mapping stays balanced
        const uint32_t g = (need + 7u) & ~7u;
        return (int)(g ? g : 8u);
    }
    return want;
}

// variant bits 4-5 select the kernel shape (benchmarks): 0 = shipped.
hipError_t launch_fill(hipStream_t s, const LaunchCfg& cfg, const LayoutView& lv, void* body,
                       uint32_t tile_begin, uint32_t tile_end)
{
    if (tile_end <= tile_begin || lv.n_obj == 0) return hipSuccess;
    if (tile_end > lv.n_tiles) tile_end = lv.n_tiles;
    const int grid = clamp_grid(cfg.grid_blocks, tile_end - tile_begin);
    const int variant = cfg.variant & 1;
    sdesc_ptr sd = (sdesc_ptr)lv.descs;
    u32x4* b = static_cast<u32x4*>(body);
#define OVAL_LAUNCH_FILL(U, P) fill_kernel<U, P><<<grid, kBlockThreads, 0, s>>>(sd, lv.tile_first, lv.n_obj, b, tile_begin, tile_end, variant)
    switch ((cfg.variant >> 4) & 3) {
    case 1: OVAL_LAUNCH_FILL(8, 1); break;
    case 2: OVAL_LAUNCH_FILL(4, 2); break;
    case 3: OVAL_LAUNCH_FILL(4, 4); break;
    default: OVAL_LAUNCH_FILL(4, 1); break;
    }
#undef OVAL_LAUNCH_FILL
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
    if (tile_end <= tile_begin || lv.n_obj == 0) return hipSuccess;
    if (tile_end > lv.n_tiles) tile_end = lv.n_tiles;
    const int grid = clamp_grid(cfg.grid_blocks, tile_end - tile_begin);
    const int variant = cfg.variant & 1;
    sdesc_ptr sd = (sdesc_ptr)lv.descs;
    const u32x4* b = static_cast<const u32x4*>(body);
#define OVAL_LAUNCH_VAL(U, P) validate_kernel<U, P><<<grid, kBlockThreads, 0, s>>>(sd, lv.tile_first, lv.n_obj, lv.span_units, b, tile_begin, tile_end, results, counters, variant)
    switch ((cfg.variant >> 4) & 3) {
    case 1: OVAL_LAUNCH_VAL(8, 1); break;
    case 2: OVAL_LAUNCH_VAL(4, 2); break;
    case 3: OVAL_LAUNCH_VAL(2, 1); break;
    default: OVAL_LAUNCH_VAL(4, 1); break;
    }
#undef OVAL_LAUNCH_VAL
    return hipGetLastError();
}

}  // namespace oval
