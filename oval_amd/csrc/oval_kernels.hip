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
// Both passes are launched as short-lived workgroups of 256 threads in address
// order (fill: one 16-byte store per thread, a workgroup per 4 KiB chunk;
// validate: four loads per thread, a wavefront per chunk).  The accesses in
// flight then cover a few MiB that sweep the body once, which this memory
// system sustains 10-17 % faster (writes) and ~3 % faster (reads) than
// persistent wavefronts that each stream a tile of their own (measured with
// tools/write_shapes.hip and with earlier persistent versions of these kernels).
//
// The walk over objects is done once per wavefront on the scalar unit: the
// chunk table names the first object of the chunk, descriptors are fetched
// with scalar loads (constant address space, one s_load_dwordx16 each), and a
// wavefront walks forward only across the object boundaries inside its range.
// The fill is sensitive to the latency of that lookup (one store per thread
// means 8 MiB of body in flight per lookup), so for all but very large objects
// it reads a per-chunk copy of the descriptor in parallel with the chunk table
// entry instead of following it.
//
// Stores are non-temporal: a fill that leaves its lines dirty in L2 / Infinity
// Cache makes the validate pass that follows pay for their write-back (measured:
// validate 6.9 -> 5.3 TB/s behind a plain-store fill).
#include <cstring>
#include <mutex>

#include "oval_kernels.h"
#include "oval_trimspace.h"

namespace oval {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4* sdesc_ptr;
typedef const __attribute__((address_space(4))) uint32_t* sidx_ptr;

constexpr uint32_t kNone = 0xFFFFFFFFu;

// bytes 4k..4k+3 of the filler, as a little-endian dword
__device__ __forceinline__ uint32_t filler_dword(uint32_t k) { return 0x03020100u + 0x04040404u * k; }

__device__ __forceinline__ uint32_t reg_byte(const u32x4& v, int i)
{
    uint32_t w = (i & 8) ? ((i & 4) ? v.w : v.z) : ((i & 4) ? v.y : v.x);
    return (w >> ((i & 3) * 8)) & 0xFFu;
}

// ---------------------------------------------------------------------------
// chunk table: chunk c (4 KiB = 16 units) -> first object whose end lies beyond
// unit 16 c, and (optionally) a copy of that object's descriptor.  A 64 KiB tile
// starts at chunk 16 t.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_chunk_table_kernel(const u32x4* __restrict__ descs,
                                                                uint32_t n_obj,
                                                                uint32_t* __restrict__ chunk_first,
                                                                u32x4* __restrict__ chunk_desc,
                                                                uint32_t n_chunks)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks) return;
    const uint32_t g = t * kUnitsPerChunk;
    uint32_t lo = 0, hi = n_obj;                  // first o in [lo,hi] with end(o) > g
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const u32x4 s3 = descs[4u * mid + 3u];    // body_offset (8) | n_units (4) | lens+flags (4)
        const uint32_t end = ((s3.x >> 8) | (s3.y << 24)) + s3.z;
        if (end > g) hi = mid; else lo = mid + 1;
    }
    chunk_first[t] = lo;
    if (chunk_desc) {
        const uint32_t oc = lo < n_obj ? lo : n_obj - 1u;
        for (int q = 0; q < 4; ++q) chunk_desc[4u * t + q] = descs[4u * oc + q];
    }
}

// ---------------------------------------------------------------------------
// Descriptor table built on the device from compact records (12 bytes per
// object: object ID, write count, units) and a short segment table (one entry
// per run of objects that share bucket, worker ID, time and key range): what
// the reference keeps per object is just that (internal/object/object.go:21-25),
// the key follows from the ID (object.go:48-50: "ov" + 10 hex digits) and
// bodies lie back to back, so the 64-byte descriptors need not cross PCIe.
// Three launches: units per block of 1024 objects; the running sum over the
// blocks (one workgroup); then every block scans its own objects and writes
// their descriptors.
// ---------------------------------------------------------------------------
constexpr uint32_t kRecsPerBlock = 1024;                      // 256 threads x 4 records

struct RecTriple { uint32_t obj_id, write_count, n_units; };

__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t* lds)
{
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wv] = v;
    __syncthreads();
    const uint32_t total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(256) void rec_block_units_kernel(const RecTriple* __restrict__ recs, uint32_t n_obj,
                                                              unsigned long long* __restrict__ block_units)
{
    __shared__ uint32_t lds[4];
    const uint32_t base = blockIdx.x * kRecsPerBlock + threadIdx.x * 4u;
    unsigned long long mine = 0;
    for (uint32_t k = 0; k < 4u; ++k)
        if (base + k < n_obj) mine += recs[base + k].n_units;
    // a block's units fit 42 bits; summed in two 32-bit halves
    const uint32_t lo = block_sum_u32((uint32_t)(mine & 0xFFFFFu), lds);              // 20 bits x 256 threads
    const uint32_t hi = block_sum_u32((uint32_t)(mine >> 20), lds);
    if (threadIdx.x == 0) block_units[blockIdx.x] = ((unsigned long long)hi << 20) + lo;
}

// block_units[b] := units in blocks 0 .. b-1 (in place), one workgroup
__global__ __launch_bounds__(1024) void scan_block_units_kernel(unsigned long long* __restrict__ block_units, uint32_t n_blocks)
{
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n_blocks ? block_units[i] : 0ull;
        part[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024u; d <<= 1) {                // inclusive scan in shared memory
            const unsigned long long add = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const unsigned long long c = carry;
        if (i < n_blocks) block_units[i] = c + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023u) carry = c + part[1023];
        __syncthreads();
    }
}

__device__ __forceinline__ uint32_t hex_pair(uint32_t byte)      // two lower-case hex digits, high digit in the low byte
{
    const uint32_t h = byte >> 4, l = byte & 15u;
    return (h + (h < 10u ? 48u : 87u)) | ((l + (l < 10u ? 48u : 87u)) << 8);
}

__global__ __launch_bounds__(256) void expand_descs_kernel(const RecTriple* __restrict__ recs, uint32_t n_obj,
                                                           const oval_obj_segment* __restrict__ segs, uint32_t n_segs,
                                                           const unsigned long long* __restrict__ block_base,
                                                           u32x4* __restrict__ descs)
{
    __shared__ unsigned long long wave_units[4];
    const uint32_t base = blockIdx.x * kRecsPerBlock + threadIdx.x * 4u;
    RecTriple r[4];
    unsigned long long mine = 0;
    for (uint32_t k = 0; k < 4u; ++k) {
        r[k] = base + k < n_obj ? recs[base + k] : RecTriple{0u, 0u, 0u};
        mine += r[k].n_units;
    }
    // exclusive scan of `mine` over the workgroup: within the wavefront by shuffles, across the four through LDS
    unsigned long long incl = mine;
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    const uint32_t wv = threadIdx.x >> 6;
    if (lane == 63) wave_units[wv] = incl;
    __syncthreads();
    unsigned long long start = block_base[blockIdx.x] + incl - mine;
    for (uint32_t w = 0; w < wv; ++w) start += wave_units[w];
    if (base >= n_obj) return;
    // the segment of this thread's first object; its others walk on from there
    uint32_t lo = 0, hi = n_segs - 1u;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if (segs[mid].first_obj <= base) lo = mid; else hi = mid - 1u;
    }
    uint32_t sg = lo;
    for (uint32_t k = 0; k < 4u && base + k < n_obj; ++k) {
        const uint32_t i = base + k;
        while (sg + 1u < n_segs && segs[sg + 1u].first_obj <= i) ++sg;
        const oval_obj_segment& s = segs[sg];
        const u32x4* sb = reinterpret_cast<const u32x4*>(s.bucket);
        const unsigned long long id = s.key_id_base + r[k].obj_id;      // below 2^40 (checked on the host)
        // "ov" + 10 hex digits, most significant first (object.go:48-50)
        const uint32_t k0 = 0x766Fu | (hex_pair((uint32_t)(id >> 32) & 0xFFu) << 16);
        const uint32_t k1 = hex_pair((uint32_t)(id >> 24) & 0xFFu) | (hex_pair((uint32_t)(id >> 16) & 0xFFu) << 16);
        const uint32_t k2 = hex_pair((uint32_t)(id >> 8) & 0xFFu) | (hex_pair((uint32_t)id & 0xFFu) << 16);
        const unsigned long long off = start * OVAL_UNIT_SIZE;
        descs[4u * i] = sb[0];
        descs[4u * i + 1u] = (u32x4){k0, k1, k2, r[k].write_count};
        descs[4u * i + 2u] = (u32x4){0u, s.worker_id, (uint32_t)s.unix_micro, (uint32_t)(s.unix_micro >> 32)};
        descs[4u * i + 3u] = (u32x4){(uint32_t)off, (uint32_t)(off >> 32), r[k].n_units,
                                     (uint32_t)s.bucket_len | (12u << 8) | ((uint32_t)s.flags << 16)};
        start += r[k].n_units;
    }
}

// ---------------------------------------------------------------------------
// Reads the part of the chunk table, of the per-chunk descriptor copies (if
// the layout has them) and of the descriptor table that the chunks
// [chunk_begin, chunk_end) refer to, so that it sits in the memory-side cache
// when the pass over those chunks starts.  Every wavefront of a pass starts
// with a lookup; with 8 MiB of body in flight per lookup latency, cold HBM
// round trips under a saturated write stream hold the fill at 3.3-4.8 TB/s
// where warm tables give 6.9-7.2 (the pre-pass itself costs 1-2 %).  The validate kernel has its body loads
// in flight while it looks up and does not need this (measured: no difference).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warm_tables_kernel(const u32x4* __restrict__ descs,
                                                          const uint32_t* __restrict__ chunk_first,
                                                          uint32_t n_obj, uint32_t chunk_begin, uint32_t chunk_end,
                                                          const u32x4* __restrict__ chunk_desc)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t acc = 0;
    // chunk_begin and chunk_end are multiples of kChunksPerTile, so the range is whole 64-byte lines
    const u32x4* cf = reinterpret_cast<const u32x4*>(chunk_first + chunk_begin);
    for (uint32_t i = tid; i < (chunk_end - chunk_begin) / 4u; i += stride) {
        const u32x4 v = cf[i];
        acc |= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (chunk_desc) {
        const u32x4* cd = chunk_desc + (size_t)chunk_begin * 4u;
        for (uint32_t i = tid; i < (chunk_end - chunk_begin) * 4u; i += stride) {
            const u32x4 v = cd[i];
            acc |= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    const uint32_t o_lo = chunk_first[chunk_begin];
    uint32_t o_hi = chunk_first[chunk_end - 1u] + kUnitsPerChunk + 1u;   // a chunk holds at most 16 objects
    if (o_hi > n_obj) o_hi = n_obj;
    if (o_lo < o_hi) {
        const u32x4* d = descs + (size_t)o_lo * 4u;
        for (uint32_t i = tid; i < (o_hi - o_lo) * 4u; i += stride) {
            const u32x4 v = d[i];
            acc |= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    asm volatile("" ::"v"(acc));                                // the loads are the point
}

// ---------------------------------------------------------------------------
// wave-uniform object walk
// ---------------------------------------------------------------------------
struct WObj {                    // the current object
    uint32_t o, start, end, lf, ubias;   // ubias = first_unit - start, so unitCount = g + ubias
    u32x4 d0, d1, d2;            // bytes [0,48) of the descriptor = header template
};

struct WRaw {                    // a fetched descriptor, not yet decoded
    uint32_t o;
    u32x4 d0, d1, d2, d3;
};

// n_obj >= 1.  The load is unconditional (index clamped to the last object);
// r.o == n_obj then says "no such object".
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
// One 16-byte store per thread; workgroup b covers units [unit_begin + b *
// THREADS/16, ...).  A wavefront (4 consecutive units) finds its object from
// the chunk table and walks forward only across the object boundaries inside
// its 1 KiB; units that belong to no object (gaps, the tail of the last tile)
// are not written.
template <int THREADS, bool ONE_LEVEL>
__global__ __launch_bounds__(THREADS) void fill_kernel(sdesc_ptr descs, sidx_ptr chunk_first, sdesc_ptr chunk_desc,
                                                       uint32_t n_obj, u32x4* __restrict__ body, uint32_t unit_begin)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};
    const uint32_t g4 = unit_begin + blockIdx.x * (THREADS / 16u) +
                        __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 4u;
    const uint32_t g = g4 + grp;
    const uint32_t c = g4 / kUnitsPerChunk;
    WRaw r;
    WObj w;
    if (ONE_LEVEL) {
        // the chunk's own copy of its first object's descriptor: no dependent second load
        const uint32_t o0 = chunk_first[c];
        r.o = o0 < n_obj ? o0 : n_obj;
        r.d0 = chunk_desc[4u * c];
        r.d1 = chunk_desc[4u * c + 1u];
        r.d2 = chunk_desc[4u * c + 2u];
        r.d3 = chunk_desc[4u * c + 3u];
    } else {
        w_fetch(descs, n_obj, chunk_first[c], r);
    }
    w_activate(r, n_obj, w);
    u32x4* const p = body + (size_t)g4 * 16u + (uint32_t)lane;
    // Each part of the wavefront stores as soon as its object is known: settling
    // all lanes first and storing once measured 2-3 % slower, a workgroup
    // barrier before the store 4-8 %.
    for (;;) {
        if (g >= w.start && g < w.end) {
            u32x4 e = w_lane_template(w, j, cpat);
            e.x += g * gmul;
            __builtin_nontemporal_store(e, p);
        }
        if (w.end >= g4 + 4u) break;                             // kNone for "no more objects"
        w_fetch(descs, n_obj, w.o + 1u, r);
        w_activate(r, n_obj, w);
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

// Per-field tallies stay in a vector register until the wavefront ends: every
// field is judged by one fixed lane of a unit, so that lane counts its own
// verdicts (at most four, one per wavefront instruction) in a 6-bit slot of
// `fields`, and the twelve header lanes are added up once (w_totals_flush).
struct WTotals {
    uint32_t units, strict, objs;   // wave-uniform
    uint32_t fields;                // per lane: bucket | key << 6 | write count << 12 | offset << 18 | worker << 24
};

// Bad-unit tallies go to one of kTallyStripes copies of the counter vector (one
// cache line each, picked by workgroup), not to the caller's counters: with
// every wavefront of a badly corrupted batch adding to the same line, the
// atomics serialise (measured: 70 ms instead of 1.3 ms for a cfg2 batch whose
// every unit is bad).  fold_tallies_kernel adds the stripes up afterwards.
// A stripe holds the eight tallies as four 64-bit words of two 32-bit counts
// each (a batch has fewer than 2^32 units), so that a wavefront adds all of
// them with ONE atomic instruction of four lanes:
//   word 0: bad objects | bad units << 32       word 1: bucket | key << 32
//   word 2: write count | offset << 32          word 3: worker ID | strict bad units << 32
__device__ __forceinline__ unsigned long long* tally_of(unsigned long long* stripes)
{
    return stripes + (size_t)(blockIdx.x & (kTallyStripes - 1u)) * kTallyStride;
}

// One lane publishes the wavefront's outcome for object w.o.  An object that
// lies wholly inside this wavefront's chunk [g0, g0 + 16) has no other writer:
// its result is stored plainly and it is counted as a bad object in the
// wavefront's own tallies.  Otherwise a 64-bit atomicMin keeps the lowest bad
// unit together with its field mask.
__device__ __forceinline__ void w_flush(oval_obj_result* __restrict__ results,
                                        unsigned long long* __restrict__ stripes, const WObj& w, WRun& r,
                                        WTotals& t, uint32_t g0, int lane)
{
    if (r.cnt != 0u || r.sfirst != kNone) {
        const bool sole = w.start >= g0 && w.end <= g0 + kUnitsPerChunk;
        if (sole) {
            if (lane == 0) {
                const uint32_t first = r.cnt != 0u ? r.first + w.ubias : kNone;
                const uint32_t mask = r.cnt != 0u ? r.mask : kNone;
                reinterpret_cast<u32x4*>(results)[w.o] = (u32x4){mask, first, r.cnt, r.sfirst + w.ubias};
            }
            t.objs += r.cnt != 0u ? 1u : 0u;
        } else if (lane == 0) {
            oval_obj_result* R = results + w.o;
            if (r.cnt != 0u) {
                const unsigned long long key =
                    ((unsigned long long)(r.first + w.ubias) << 32) | (unsigned long long)r.mask;
                atomicMin(reinterpret_cast<unsigned long long*>(R), key);
                const uint32_t old = atomicAdd(&R->bad_units, r.cnt);
                if (old == 0u && stripes) atomicAdd(&tally_of(stripes)[0], 1ull);    // a bad object (word 0, low half)
            }
            atomicMin(&R->strict_first_bad_unit, r.sfirst + w.ubias);
        }
        r.first = kNone; r.mask = 0; r.cnt = 0; r.sfirst = kNone;
    }
}

// Adds a wavefront's bad-unit tallies to its workgroup's stripe (nothing to do
// for a clean wavefront: every reference-visible mismatch is also a strict one).
__device__ __forceinline__ void w_totals_flush(unsigned long long* __restrict__ stripes, const WTotals& tot,
                                               int lane)
{
    if (!stripes || tot.strict == 0u) return;
    // The header lanes 16 g + 0 / 1 / 2 hold disjoint slots (bucket; key, write count; offset, worker): at most 16
    // per slot over the wavefront.  Summed within each quad by two DPP steps, then over the four groups.
    const int f0 = (lane & 15) < 3 ? (int)tot.fields : 0;
    const int f1 = f0 + __builtin_amdgcn_update_dpp(0, f0, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
    const int f2 = f1 + __builtin_amdgcn_update_dpp(0, f1, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
    const uint32_t f = (uint32_t)__builtin_amdgcn_readlane(f2, 0) + (uint32_t)__builtin_amdgcn_readlane(f2, 16) +
                       (uint32_t)__builtin_amdgcn_readlane(f2, 32) + (uint32_t)__builtin_amdgcn_readlane(f2, 48);
    // lane k < 4 adds word k of the stripe: slots 0, 12, 24 of f are the low halves of words 1-3, slots 6 and 18
    // the high halves of words 1 and 2
    const uint32_t sh = 12u * (uint32_t)((lane - 1) & 3);       // (lanes 1-3: 0, 12, 24; the other lanes' values are not used)
    const uint32_t lo = lane == 0 ? tot.objs : (f >> (sh & 31u)) & 63u;
    const uint32_t hi = lane == 0 ? tot.units : lane == 3 ? tot.strict : (f >> ((sh + 6u) & 31u)) & 63u;
    if (lane < 4 && (lo | hi) != 0u)
        atomicAdd(&tally_of(stripes)[lane], (unsigned long long)lo | ((unsigned long long)hi << 32));
}

// Exact per-field outcome of one lane's 16 bytes (pattern.go:140-170) for the
// lanes flagged `bad`: those that failed the 128-bit compare or whose descriptor
// asks for the exact check.  Every field is judged by one fixed lane of a unit's
// sixteen, and no lane judges more than two: the result is two bits,
//   lane 0: A = bucket            lane 1: A = key, B = write count
//   lane 2: A = offset, B = worker ID        lanes 3-15: A = filler
// The numeric fields are compares of the XOR of the lane's bytes with its
// template, selected by the lane's place without a branch; the TrimSpace compare
// of a name field sits behind a wave-uniform branch that is taken only when some
// flagged name lane's bytes differ from the formatted expected name (or its
// expected name has white space at an edge, `force`): a stale generation, where
// only the write count differs, never takes it.  Called by all 64 lanes.
__device__ __forceinline__ uint32_t lane_verdicts(const u32x4& v, uint32_t ex, const u32x4& tmpl, uint32_t lf,
                                                  uint32_t force, bool bad, int j)
{
    const uint32_t dx = v.x ^ ex, dy = v.y ^ tmpl.y, dz = v.z ^ tmpl.z, dw = v.w ^ tmpl.w;
    uint32_t a = (j >= 3) ? 1u : (j == 2 && dx != 0u) ? 1u : 0u;       // a flagged filler lane is flagged because its bytes differ
    const uint32_t b = (j == 2) ? ((dy | force) != 0u ? 2u : 0u)       // force: worker ID outside uint32
                                : (j == 1 && dw != 0u) ? 2u : 0u;
    const bool name = bad && j < 2 && (dx | dy | dz | (j == 0 ? dw : 0u) | force) != 0u;   // bucket: 16 bytes, key: the first 12
    if (__any(name)) {
        if (name) {
            const int width = j == 0 ? 16 : 12;
            const int len = (int)((j == 0 ? lf : lf >> 8) & 0xFFu);
            if (!trimmed_equals([&](int i) { return reg_byte(v, i); }, width, [&](int i) { return reg_byte(tmpl, i); }, len))
                a = 1u;
        }
    }
    return bad ? (a | b) : 0u;
}

// Where a lane's two verdicts are counted in WTotals::fields (see there): shift of A | shift of B << 8.
__device__ __forceinline__ uint32_t lane_field_slots(int j)
{
    return j == 0 ? 0u : j == 1 ? (6u | 12u << 8) : j == 2 ? (18u | 24u << 8) : 0u;   // (filler lanes: never read back)
}

// Exact verdict for the lanes flagged `bad` in one wavefront instruction (four
// units, the flagged ones all inside object w).  Two ballots put the verdicts of
// all four units into scalar registers, A and B of lane l of unit u at bit
// 16 u + l; shifted down to bit 16 u the five reference fields combine with
// scalar ORs: no cross-lane traffic.  The field mask of a unit is picked out of
// the ballots only for the first bad unit of an object.  (Six ballots, one per
// field, cost a batch whose every unit is bad 5 % more; counting the fields with
// scalar population counts of the ballots instead of in a vector register cost 3 %.)
__device__ __forceinline__ void w_handle_bad(const u32x4& v, uint32_t ex, const u32x4& tmpl, uint32_t lf,
                                             uint32_t force, bool bad, int j, uint32_t slots, uint32_t g4, WRun& r,
                                             WTotals& t)
{
    const uint32_t ab = lane_verdicts(v, ex, tmpl, lf, force, bad, j);
    const unsigned long long A = __ballot((ab & 1u) != 0u), B = __ballot((ab & 2u) != 0u);
    constexpr unsigned long long m0 = 0x0001000100010001ull, low15 = 0x7FFF7FFF7FFF7FFFull;
    // bit 16 u: unit u has a reference-visible mismatch (A of lanes 0-2, B of lanes 1-2)
    const unsigned long long refu = (A | (A >> 1) | (A >> 2) | (B >> 1) | (B >> 2)) & m0;
    // bit 16 u: some lane of unit u is set at all, filler included (a carry out of the group's low 15 bits, or bit 15 itself)
    const unsigned long long X = A | B;
    const unsigned long long anyu = ((((X & low15) + low15) | X) >> 15) & m0;
    // this lane's own verdicts, each in its field's slot
    t.fields += ((ab & 1u) << (slots & 31u)) + ((ab >> 1) << (slots >> 8));
    if (refu) {
        const uint32_t sh = (uint32_t)__builtin_ctzll(refu);    // 16 x the lowest bad unit
        if (r.cnt == 0u) {                                      // units arrive in ascending order
            r.first = g4 + (sh >> 4);
            const uint32_t a3 = (uint32_t)(A >> sh) & 7u, b3 = (uint32_t)(B >> sh) & 6u;   // bucket, key, offset | write count, worker
            static_assert(OVAL_BAD_BUCKET == 1u && OVAL_BAD_KEY == 2u && OVAL_BAD_WRITECNT == 4u && OVAL_BAD_OFFSET == 8u &&
                              OVAL_BAD_WORKER == 16u, "the field mask is assembled by shifts");
            r.mask = (a3 & 3u) | ((b3 & 2u) << 1) | ((a3 & 4u) << 1) | ((b3 & 4u) << 2);
        }
        const uint32_t n = (uint32_t)__builtin_popcountll(refu);
        r.cnt += n;
        t.units += n;
    }
    if (anyu) {
        if (r.sfirst == kNone) r.sfirst = g4 + ((uint32_t)__builtin_ctzll(anyu) >> 4);
        t.strict += (uint32_t)__builtin_popcountll(anyu);
    }
}

__device__ __forceinline__ uint32_t w_diff(const u32x4& v, uint32_t ex, const u32x4& tmpl, uint32_t zwmask)
{
    return (v.x ^ ex) | (v.y ^ tmpl.y) | (((v.z ^ tmpl.z) | (v.w ^ tmpl.w)) & zwmask);
}

// The object walk of validate_kernel (below) as a function, for the small-object
// kernel's wavefronts that need it: over the chunk [g0, g0 + 16) whose sixty-four
// 16-byte pieces the wavefront holds in v, from object o0 on.  (validate_kernel
// keeps its own copy of these lines: as a call the register allocator spills there.)
__device__ __forceinline__ void validate_chunk_walk(sdesc_ptr descs, uint32_t n_obj, uint32_t o0, uint32_t g0,
                                                    const u32x4 (&v)[4], oval_obj_result* __restrict__ results,
                                                    unsigned long long* __restrict__ stripes, int lane)
{
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const uint32_t zwmask = (j == 2) ? 0u : 0xFFFFFFFFu;       // pattern.go:172 "Skip the unix time area."
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};
    WRaw r;
    WObj w;
    w_fetch(descs, n_obj, o0, r);
    const WRaw r0 = r;                                         // kept for the second walk of a bad wavefront
    w_activate(r, n_obj, w);
    u32x4 tmpl = w_lane_template(w, j, cpat);
    uint32_t force = w_lane_force(w, j);
    uint32_t anybad = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t g4 = g0 + 4u * i;
        const uint32_t g = g4 + grp;
        for (;;) {
            if (g >= w.start && g < w.end) anybad |= w_diff(v[i], tmpl.x + g * gmul, tmpl, zwmask) | force;
            if (w.end >= g4 + 4u) break;
            w_fetch(descs, n_obj, w.o + 1u, r);
            w_activate(r, n_obj, w);
            tmpl = w_lane_template(w, j, cpat);
            force = w_lane_force(w, j);
        }
        // a wavefront that has seen a mismatch walks again from the start anyway: no need to finish this walk
        if (i < 3 && __any(anybad != 0u)) break;
    }
    if (!__any(anybad != 0u)) return;

    w_activate(r0, n_obj, w);
    tmpl = w_lane_template(w, j, cpat);
    force = w_lane_force(w, j);
    const uint32_t slots = lane_field_slots(j);
    WRun run = {kNone, 0, 0, kNone};
    WTotals tot = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t g4 = g0 + 4u * i;
        const uint32_t g = g4 + grp;
        const u32x4 vi = v[i];
        for (;;) {
            const uint32_t ex = tmpl.x + g * gmul;
            const bool bad = (g >= w.start && g < w.end) && ((w_diff(vi, ex, tmpl, zwmask) | force) != 0u);
            if (__any(bad)) w_handle_bad(vi, ex, tmpl, w.lf, force, bad, j, slots, g4, run, tot);
            if (w.end >= g4 + 4u) break;
            w_flush(results, stripes, w, run, tot, g0, lane);
            w_fetch(descs, n_obj, w.o + 1u, r);
            w_activate(r, n_obj, w);
            tmpl = w_lane_template(w, j, cpat);
            force = w_lane_force(w, j);
        }
    }
    if (w.o < n_obj) w_flush(results, stripes, w, run, tot, g0, lane);
    w_totals_flush(stripes, tot, lane);
}

// Four loads per thread: a wavefront takes one whole 4 KiB chunk (wave-
// contiguous), issues its loads, looks its object up while they are in flight,
// then compares; only a wavefront that sees a mismatch runs the exact verdicts
// (on the registers it already holds).
template <int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void validate_kernel(
    sdesc_ptr descs, sidx_ptr chunk_first, uint32_t n_obj, uint32_t limit_units, const u32x4* __restrict__ body,
    uint32_t chunk_begin, oval_obj_result* __restrict__ results, unsigned long long* __restrict__ stripes)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const uint32_t zwmask = (j == 2) ? 0u : 0xFFFFFFFFu;       // pattern.go:172 "Skip the unix time area."
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};
    const uint32_t chunk = chunk_begin + blockIdx.x * (THREADS / 64u) +
                           __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g0 = chunk * kUnitsPerChunk;
    const u32x4* const p = body + (size_t)g0 * 16u + (uint32_t)lane;
    u32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = (u32x4){0u, 0u, 0u, 0u};
        if (g0 + 4u * i + grp < limit_units) v[i] = __builtin_nontemporal_load(p + (size_t)i * 64u);
    }
    const uint32_t o0 = chunk_first[chunk];
    WRaw r;
    WObj w;
    w_fetch(descs, n_obj, o0, r);
    const WRaw r0 = r;                                         // kept for the second walk of a bad wavefront
    w_activate(r, n_obj, w);
    u32x4 tmpl = w_lane_template(w, j, cpat);
    uint32_t force = w_lane_force(w, j);
    uint32_t anybad = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t g4 = g0 + 4u * i;
        const uint32_t g = g4 + grp;
        for (;;) {
            if (g >= w.start && g < w.end) anybad |= w_diff(v[i], tmpl.x + g * gmul, tmpl, zwmask) | force;
            if (w.end >= g4 + 4u) break;
            w_fetch(descs, n_obj, w.o + 1u, r);
            w_activate(r, n_obj, w);
            tmpl = w_lane_template(w, j, cpat);
            force = w_lane_force(w, j);
        }
        // a wavefront that has seen a mismatch walks again from the start anyway: no need to finish this walk
        if (i < 3 && __any(anybad != 0u)) break;
    }
    if (!__any(anybad != 0u)) return;

    w_activate(r0, n_obj, w);
    tmpl = w_lane_template(w, j, cpat);
    force = w_lane_force(w, j);
    const uint32_t slots = lane_field_slots(j);
    WRun run = {kNone, 0, 0, kNone};
    WTotals tot = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t g4 = g0 + 4u * i;
        const uint32_t g = g4 + grp;
        const u32x4 vi = v[i];
        for (;;) {
            const uint32_t ex = tmpl.x + g * gmul;
            const bool bad = (g >= w.start && g < w.end) && ((w_diff(vi, ex, tmpl, zwmask) | force) != 0u);
            if (__any(bad)) w_handle_bad(vi, ex, tmpl, w.lf, force, bad, j, slots, g4, run, tot);
            if (w.end >= g4 + 4u) break;
            w_flush(results, stripes, w, run, tot, g0, lane);
            w_fetch(descs, n_obj, w.o + 1u, r);
            w_activate(r, n_obj, w);
            tmpl = w_lane_template(w, j, cpat);
            force = w_lane_force(w, j);
        }
    }
    if (w.o < n_obj) w_flush(results, stripes, w, run, tot, g0, lane);
    w_totals_flush(stripes, tot, lane);
}

// ---------------------------------------------------------------------------
// Objects of a few units (the reference's smallest object is one unit,
// pattern.go:23-34).  The walk above meets a new object every unit or two and
// pays a dependent scalar load for each: up to 15 in a row for a chunk of
// one-unit objects.  Layouts whose objects average under a few KiB take these
// kernels instead.  Objects hold at least one unit each (zero-unit objects
// aside), so the objects that meet a chunk are among the sixteen that follow
// its first one in the table: lane j of every 16-lane group loads the extent of
// candidate o0 + j with ONE vector load, each group finds its unit's object by a
// ballot over the candidates' extents, and the header lanes load their 16-byte
// slice of that object's descriptor: two dependent vector loads however many
// objects the chunk holds.  A wavefront whose sixteen candidates do not reach
// the end of its units (zero-unit objects in between) takes the scalar walk.
// ---------------------------------------------------------------------------
struct SmallCand {
    uint32_t start, end, lf;     // this lane's candidate: extent in units, lens and flags; start = end = kNone if none
    bool more;                   // lane 15 of a group: objects follow the sixteenth candidate
};

__device__ __forceinline__ SmallCand small_candidates(const u32x4* __restrict__ gdescs, uint32_t n_obj, uint32_t o0, int j)
{
    SmallCand c;
    const uint32_t o = o0 + (uint32_t)j;
    const uint32_t oc = o < n_obj ? o : n_obj - 1u;            // n_obj >= 1: the load is unconditional
    const u32x4 d3 = gdescs[4u * oc + 3u];
    const uint32_t start = (d3.x >> 8) | (d3.y << 24);
    c.start = o < n_obj ? start : kNone;
    c.end = o < n_obj ? start + d3.z : kNone;
    c.lf = d3.w;
    c.more = o + 1u < n_obj;
    return c;
}

// Candidate index (0-15) whose extent holds unit g of this lane's group, from the ballot of the group's
// sixteen candidate lanes; `found` says whether there is one (a unit in a gap belongs to no object).
__device__ __forceinline__ uint32_t small_match(const SmallCand& c, uint32_t g, uint32_t grp, bool& found)
{
    const unsigned long long m = __ballot(g >= c.start && g < c.end);
    const uint32_t bits = (uint32_t)(m >> (16u * grp)) & 0xFFFFu;
    found = bits != 0u;
    return (uint32_t)__builtin_ctz(bits | 0x10000u);
}

// A wavefront fills one whole chunk (four stores per thread): with three dependent
// loads in front of the first store, 4 KiB in flight per wavefront hide that
// latency better than the 1 KiB of fill_kernel's shape.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void fill_small_kernel(sdesc_ptr descs, const u32x4* __restrict__ gdescs,
                                                             sidx_ptr chunk_first, uint32_t n_obj,
                                                             u32x4* __restrict__ body, uint32_t chunk_begin)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};
    const uint32_t chunk = chunk_begin + blockIdx.x * (THREADS / 64u) +
                           __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g0 = chunk * kUnitsPerChunk;
    const uint32_t o0 = chunk_first[chunk];
    if (o0 >= n_obj) return;                                     // nothing ends beyond this chunk's first unit
    u32x4* const p = body + (size_t)g0 * 16u + (uint32_t)lane;
    const SmallCand c = small_candidates(gdescs, n_obj, o0, j);
    // units beyond the sixteenth candidate's end may belong to objects this wavefront has not looked at
    if (__any(j == 15 && c.more && c.end < g0 + kUnitsPerChunk)) {
        const uint32_t gmul = (j == 2) ? 256u : 0u;
        WRaw r;
        WObj w;
        w_fetch(descs, n_obj, o0, r);
        w_activate(r, n_obj, w);
#pragma unroll 1
        for (uint32_t i = 0; i < 4u; ++i) {
            const uint32_t g4 = g0 + 4u * i;
            const uint32_t g = g4 + grp;
            for (;;) {
                if (g >= w.start && g < w.end) {
                    u32x4 e = w_lane_template(w, j, cpat);
                    e.x += g * gmul;
                    __builtin_nontemporal_store(e, p + (size_t)i * 64u);
                }
                if (w.end >= g4 + 4u) break;                     // kNone for "no more objects"
                w_fetch(descs, n_obj, w.o + 1u, r);
                w_activate(r, n_obj, w);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t g = g0 + 4u * i + grp;
        bool found;
        const uint32_t idx = small_match(c, g, grp, found);
        const uint32_t start = (uint32_t)__shfl((int)c.start, (int)(16u * grp + (idx & 15u)));
        if (found) {
            u32x4 e = cpat;
            if (j < 3) e = gdescs[4u * (o0 + idx) + (uint32_t)j];
            if (j == 2) e.x = (e.x + (g - start)) << 8;          // uint32(unitCount * 256), unitCount = first_unit + g - start
            __builtin_nontemporal_store(e, p + (size_t)i * 64u);
        }
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void validate_small_kernel(
    sdesc_ptr descs, const u32x4* __restrict__ gdescs, sidx_ptr chunk_first, uint32_t n_obj, uint32_t limit_units,
    const u32x4* __restrict__ body, uint32_t chunk_begin, oval_obj_result* __restrict__ results,
    unsigned long long* __restrict__ stripes)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t chunk = chunk_begin + blockIdx.x * (THREADS / 64u) +
                           __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g0 = chunk * kUnitsPerChunk;
    const u32x4* const p = body + (size_t)g0 * 16u + (uint32_t)lane;
    u32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = (u32x4){0u, 0u, 0u, 0u};
        if (g0 + 4u * i + grp < limit_units) v[i] = __builtin_nontemporal_load(p + (size_t)i * 64u);
    }
    const uint32_t o0 = chunk_first[chunk];
    if (o0 >= n_obj) return;                                     // no object reaches into this chunk
    const SmallCand c = small_candidates(gdescs, n_obj, o0, j);
    if (!__any(j == 15 && c.more && c.end < g0 + kUnitsPerChunk)) {
        const uint32_t zwmask = (j == 2) ? 0u : 0xFFFFFFFFu;   // pattern.go:172 "Skip the unix time area."
        const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                            filler_dword(4u * j + 3u)};
        uint32_t anybad = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t g = g0 + 4u * i + grp;
            bool found;
            const uint32_t idx = small_match(c, g, grp, found);
            const int src = (int)(16u * grp + (idx & 15u));
            const uint32_t start = (uint32_t)__shfl((int)c.start, src);
            const uint32_t lf = (uint32_t)__shfl((int)c.lf, src);
            u32x4 t = cpat;
            if (found && j < 3) t = gdescs[4u * (o0 + idx) + (uint32_t)j];
            const uint32_t ex = (j == 2) ? (t.x + (g - start)) << 8 : t.x;
            const uint32_t force = (j < 3) ? ((lf >> 16) & (1u << j)) : 0u;
            if (found) anybad |= w_diff(v[i], ex, t, zwmask) | force;
        }
        if (!__any(anybad != 0u)) return;
    }
    // a mismatch, a descriptor that demands the exact check, or candidates that do not reach: the walk decides
    validate_chunk_walk(descs, n_obj, o0, g0, v, results, stripes, lane);
}

// ---------------------------------------------------------------------------
// Objects far larger than a workgroup's range (multipart-sized bodies).  A read
// stream is served fastest here as ONE load per thread from workgroups of 1024
// threads in address order (7.5 TB/s against 7.2 with four loads per thread,
// tools/write_shapes.hip), which leaves a wavefront 1 KiB to pay its lookup
// with: too little when every wavefront looks up for itself.  With objects of
// megabytes nearly every workgroup's 16 KiB lie inside one object, so wavefront
// 0 looks up once, leaves the sixteen lanes' expected bytes in LDS, and the
// other fifteen wavefronts compare against those.  A workgroup whose range holds
// an object boundary, a gap or a flagged descriptor, and any wavefront that sees
// a mismatch, takes the walk below over its own four units.
// ---------------------------------------------------------------------------
// validate_chunk_walk for the four units [g4, g4 + 4) of one load: one walk with
// the exact verdicts where a lane differs (a wavefront comes here on suspicion).
__device__ __forceinline__ void validate_piece_walk(sdesc_ptr descs, uint32_t n_obj, uint32_t o0, uint32_t g4,
                                                    const u32x4& v, oval_obj_result* __restrict__ results,
                                                    unsigned long long* __restrict__ stripes, int lane)
{
    const int j = lane & 15;
    const uint32_t g = g4 + ((uint32_t)lane >> 4);
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const uint32_t zwmask = (j == 2) ? 0u : 0xFFFFFFFFu;       // pattern.go:172 "Skip the unix time area."
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};
    const uint32_t slots = lane_field_slots(j);
    WRaw r;
    WObj w;
    w_fetch(descs, n_obj, o0, r);
    w_activate(r, n_obj, w);
    u32x4 tmpl = w_lane_template(w, j, cpat);
    uint32_t force = w_lane_force(w, j);
    WRun run = {kNone, 0, 0, kNone};
    WTotals tot = {0, 0, 0, 0};
    for (;;) {
        const uint32_t ex = tmpl.x + g * gmul;
        const bool bad = (g >= w.start && g < w.end) && ((w_diff(v, ex, tmpl, zwmask) | force) != 0u);
        if (__any(bad)) w_handle_bad(v, ex, tmpl, w.lf, force, bad, j, slots, g4, run, tot);
        if (w.end >= g4 + 4u) break;                             // kNone for "no more objects"
        w_flush(results, stripes, w, run, tot, kNone, lane);     // (kNone: other wavefronts may hold units of the object too)
        w_fetch(descs, n_obj, w.o + 1u, r);
        w_activate(r, n_obj, w);
        tmpl = w_lane_template(w, j, cpat);
        force = w_lane_force(w, j);
    }
    if (w.o < n_obj) w_flush(results, stripes, w, run, tot, kNone, lane);
    w_totals_flush(stripes, tot, lane);
}

constexpr uint32_t kBigThreads = 1024;                           // 16 wavefronts x 4 units = 64 units = 16 KiB = four chunks

__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void validate_big_kernel(
    sdesc_ptr descs, sidx_ptr chunk_first, uint32_t n_obj, uint32_t limit_units, const u32x4* __restrict__ body,
    uint32_t chunk_begin, oval_obj_result* __restrict__ results, unsigned long long* __restrict__ stripes)
{
    __shared__ u32x4 s_tmpl[16];
    __shared__ uint32_t s_one;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = (chunk_begin + blockIdx.x * 4u) * kUnitsPerChunk;     // the workgroup's first unit
    const uint32_t g4 = gw + 4u * wv;
    const uint32_t g = g4 + ((uint32_t)lane >> 4);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (g < limit_units) v = __builtin_nontemporal_load(body + (size_t)g4 * 16u + (uint32_t)lane);
    if (wv == 0u) {
        const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                            filler_dword(4u * j + 3u)};
        WRaw r;
        WObj w;
        w_fetch(descs, n_obj, chunk_first[gw / kUnitsPerChunk], r);
        w_activate(r, n_obj, w);
        if (lane < 16) s_tmpl[lane] = w_lane_template(w, j, cpat);
        // one object, without a flag that demands the exact check, holds every unit of the range
        if (lane == 0) s_one = (w.start <= gw && w.end >= gw + kBigThreads / 16u && (w.lf >> 16) == 0u) ? 1u : 0u;
    }
    __syncthreads();
    if (s_one != 0u) {
        const u32x4 t = s_tmpl[j];
        const uint32_t gmul = (j == 2) ? 256u : 0u;
        const uint32_t zwmask = (j == 2) ? 0u : 0xFFFFFFFFu;   // pattern.go:172 "Skip the unix time area."
        if (!__any(w_diff(v, t.x + g * gmul, t, zwmask) != 0u)) return;
    }
    validate_piece_walk(descs, n_obj, chunk_first[g4 / kUnitsPerChunk], g4, v, results, stripes, lane);
}

// ---------------------------------------------------------------------------
// Small batches whose bodies live in pinned host memory (the micro-batched
// single-object calls of the host mirror: Generate / Valid one 4-16 KiB object
// per call, internal/runner/worker.go:64-200).  One launch does everything:
// workgroup b takes job b (validate jobs first, then fill jobs), reads its
// 64-byte descriptor from host memory, and streams the body over PCIe directly
// (loads from / non-temporal stores to the pinned buffer), so a batch costs one
// kernel launch instead of copies + table builds + two launches + copies.  A
// validate job is owned by one workgroup, so its outcome needs no atomics and
// no initialisation: the four wavefronts combine in LDS and one thread stores
// the 16-byte result to host memory.  Descriptor offsets are byte offsets into
// `up` (validate) or `down` (fill).
// ---------------------------------------------------------------------------
// Completion without an event: when `done_host` is given, every workgroup makes
// its stores to host memory visible system-wide, counts itself in `done_count`
// (device memory, zero between launches), and the last one to do so writes
// `done_seq` to `done_host` (pinned).  The host then learns of the end of the
// batch by reading its own memory, a few microseconds before the stream's
// completion signal would have told it, and records no event per batch.
__device__ __forceinline__ void host_batch_done(uint32_t* done_count, uint32_t* done_host, uint32_t done_seq)
{
    if (!done_host) return;
    __threadfence_system();                                    // this thread's stores, host memory included
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t before = atomicAdd(done_count, 1u);
        if (before == gridDim.x - 1u) {
            __threadfence_system();                            // the other workgroups' stores come before the flag
            __hip_atomic_store(done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done_host, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The work of one workgroup of a host batch: job `job` (a validate job if
// is_val), described by `r`.
__device__ __forceinline__ void host_batch_job(const WRaw& r, bool is_val, uint32_t job, const u32x4* __restrict__ up,
                                               u32x4* __restrict__ down, u32x4* __restrict__ results,
                                               uint32_t* done_count, uint32_t* done_host, uint32_t done_seq);

__global__ __launch_bounds__(256) void host_batch_kernel(sdesc_ptr vdescs, sdesc_ptr fdescs, uint32_t n_val,
                                                         const u32x4* __restrict__ up, u32x4* __restrict__ down,
                                                         u32x4* __restrict__ results, uint32_t* done_count,
                                                         uint32_t* done_host, uint32_t done_seq)
{
    const bool is_val = blockIdx.x < n_val;
    const uint32_t job = is_val ? blockIdx.x : blockIdx.x - n_val;
    sdesc_ptr dp = (is_val ? vdescs : fdescs) + 4u * job;
    WRaw r;
    r.o = 0;
    r.d0 = dp[0]; r.d1 = dp[1]; r.d2 = dp[2]; r.d3 = dp[3];
    host_batch_job(r, is_val, job, up, down, results, done_count, done_host, done_seq);
}

// A batch of one job, its descriptor in the kernel arguments: one trip over
// PCIe less before the body can be touched (a call that launches on its own
// waits for every one of them).
__global__ __launch_bounds__(256) void host_one_kernel(u32x4 d0, u32x4 d1, u32x4 d2, u32x4 d3, uint32_t is_val,
                                                       const u32x4* __restrict__ up, u32x4* __restrict__ down,
                                                       u32x4* __restrict__ results, uint32_t* done_count,
                                                       uint32_t* done_host, uint32_t done_seq)
{
    WRaw r;
    r.o = 0;
    r.d0 = d0; r.d1 = d1; r.d2 = d2; r.d3 = d3;
    host_batch_job(r, is_val != 0u, 0u, up, down, results, done_count, done_host, done_seq);
}

// A batch of up to kArgJobs jobs, its descriptors in the kernel arguments
// (validate jobs first, then fill jobs): the same trip saved for every workgroup.
constexpr uint32_t kArgJobs = 32;
struct ArgDescs { u32x4 q[kArgJobs * 4u]; };                    // 2 KiB of the 4 KiB a launch may carry (48 jobs measured no better)

__global__ __launch_bounds__(256) void host_args_kernel(ArgDescs descs, uint32_t n_val, const u32x4* __restrict__ up,
                                                        u32x4* __restrict__ down, u32x4* __restrict__ results,
                                                        uint32_t* done_count, uint32_t* done_host, uint32_t done_seq)
{
    const bool is_val = blockIdx.x < n_val;
    const uint32_t job = is_val ? blockIdx.x : blockIdx.x - n_val;
    WRaw r;
    r.o = 0;
    r.d0 = descs.q[4u * blockIdx.x]; r.d1 = descs.q[4u * blockIdx.x + 1u];
    r.d2 = descs.q[4u * blockIdx.x + 2u]; r.d3 = descs.q[4u * blockIdx.x + 3u];
    host_batch_job(r, is_val, job, up, down, results, done_count, done_host, done_seq);
}

__device__ __forceinline__ void host_batch_job(const WRaw& r, bool is_val, uint32_t job, const u32x4* __restrict__ up,
                                               u32x4* __restrict__ down, u32x4* __restrict__ results,
                                               uint32_t* done_count, uint32_t* done_host, uint32_t done_seq)
{
    __shared__ uint32_t part[4][4];                            // per wavefront: first, mask, cnt, sfirst
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const uint32_t grp = (uint32_t)lane >> 4;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gmul = (j == 2) ? 256u : 0u;
    const u32x4 cpat = {filler_dword(4u * j), filler_dword(4u * j + 1u), filler_dword(4u * j + 2u),
                        filler_dword(4u * j + 3u)};
    WObj w;
    w_activate(r, 1u, w);
    const u32x4 tmpl = w_lane_template(w, j, cpat);
    if (!is_val) {
        // wavefront wv writes units start + 4 wv + 16 k ... : 1 KiB per store instruction
        for (uint32_t g4 = w.start + 4u * wv; g4 < w.end; g4 += 16u) {
            const uint32_t g = g4 + grp;
            if (g < w.end) {
                u32x4 e = tmpl;
                e.x += g * gmul;
                __builtin_nontemporal_store(e, down + (size_t)g4 * 16u + (uint32_t)lane);
            }
        }
        host_batch_done(done_count, done_host, done_seq);
        return;
    }
    const uint32_t zwmask = (j == 2) ? 0u : 0xFFFFFFFFu;       // pattern.go:172 "Skip the unix time area."
    const uint32_t force = w_lane_force(w, j);
    const uint32_t slots = lane_field_slots(j);
    WRun run = {kNone, 0, 0, kNone};
    WTotals tot = {0, 0, 0, 0};
    // 64 units (16 KiB) per round of the workgroup: four loads per thread in flight
    for (uint32_t base = w.start; base < w.end; base += 64u) {
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t g4 = base + 16u * i + 4u * wv;
            v[i] = (u32x4){0u, 0u, 0u, 0u};
            if (g4 + grp < w.end) v[i] = __builtin_nontemporal_load(up + (size_t)g4 * 16u + (uint32_t)lane);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t g4 = base + 16u * i + 4u * wv;
            const uint32_t g = g4 + grp;
            const uint32_t ex = tmpl.x + g * gmul;
            const bool bad = g < w.end && ((w_diff(v[i], ex, tmpl, zwmask) | force) != 0u);
            if (__any(bad)) w_handle_bad(v[i], ex, tmpl, w.lf, force, bad, j, slots, g4, run, tot);
        }
    }
    if (lane == 0) {
        part[wv][0] = run.first; part[wv][1] = run.mask; part[wv][2] = run.cnt; part[wv][3] = run.sfirst;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t first = kNone, mask = kNone, cnt = 0, sfirst = kNone;
        for (int k = 0; k < 4; ++k) {
            if (part[k][2] != 0u && part[k][0] < first) { first = part[k][0]; mask = part[k][1]; }
            cnt += part[k][2];
            if (part[k][3] < sfirst) sfirst = part[k][3];
        }
        results[job] = (u32x4){mask, first != kNone ? first + w.ubias : kNone, cnt,
                               sfirst != kNone ? sfirst + w.ubias : kNone};
    }
    host_batch_done(done_count, done_host, done_seq);
}

// counters += sum of the stripes; the stripes return to zero for the next pass.
__global__ __launch_bounds__(256) void fold_tallies_kernel(unsigned long long* __restrict__ stripes,
                                                           unsigned long long* __restrict__ counters)
{
    static_assert(kTallyStripes == 256, "one thread per stripe");
    __shared__ unsigned long long sum[OVAL_NUM_COUNTERS];
    if (threadIdx.x < OVAL_NUM_COUNTERS) sum[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long* mine = stripes + (size_t)threadIdx.x * kTallyStride;
    // which counters the two halves of a stripe's four words are (see tally_of)
    const int lo_of[4] = {OVAL_CNT_OBJECTS_BAD, OVAL_CNT_BAD_BUCKET, OVAL_CNT_BAD_WRITECNT, OVAL_CNT_BAD_WORKER};
    const int hi_of[4] = {OVAL_CNT_UNITS_BAD, OVAL_CNT_BAD_KEY, OVAL_CNT_BAD_OFFSET, OVAL_CNT_STRICT_UNITS_BAD};
    for (int w = 0; w < 4; ++w) {
        const unsigned long long x = mine[w];
        if (x) {
            if (x & 0xFFFFFFFFull) atomicAdd(&sum[lo_of[w]], x & 0xFFFFFFFFull);
            if (x >> 32) atomicAdd(&sum[hi_of[w]], x >> 32);
            mine[w] = 0ull;
        }
    }
    __syncthreads();
    if (threadIdx.x < OVAL_NUM_COUNTERS && sum[threadIdx.x]) atomicAdd(&counters[threadIdx.x], sum[threadIdx.x]);
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

// table[i] = results[i].first_bad_unit: one rank's slice of the job-wide first-bad-unit table
__global__ __launch_bounds__(256) void gather_first_bad_kernel(const u32x4* __restrict__ results, uint32_t n_obj,
                                                               uint32_t* __restrict__ table)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_obj; i += stride) table[i] = results[i].y;
}

// The same for a caller that does not know the sum of n_units: it is taken
// from the descriptor table here (one atomicAdd per wavefront).
__global__ __launch_bounds__(256) void init_results_sum_kernel(u32x4* __restrict__ results,
                                                               const u32x4* __restrict__ descs, uint32_t n_obj,
                                                               unsigned long long* __restrict__ counters)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long units = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_obj; i += stride) {
        results[i] = (u32x4){kNone, kNone, 0u, kNone};
        units += descs[4u * i + 3u].z;
    }
    if (!counters) return;
    for (int d = 32; d > 0; d >>= 1) units += __shfl_xor(units, d);
    if ((threadIdx.x & 63) == 0 && units) atomicAdd(&counters[OVAL_CNT_UNITS_CHECKED], units);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters[OVAL_CNT_OBJECTS_CHECKED], (unsigned long long)n_obj);
}

// ---------------------------------------------------------------------------
// Scratch of the entry point that allocates nothing (launch_validate_noalloc):
// the striped tallies followed by the chunk table of the largest body it
// accepts.  One copy per GPU the library runs on, held until the process ends.
// ---------------------------------------------------------------------------
constexpr uint32_t kScratchTallyWords = (uint32_t)(kTallyBytes / 4);
constexpr uint32_t kNoAllocMaxChunks = kNoAllocMaxTiles * kChunksPerTile;
static __device__ __align__(256) uint32_t scratch[kScratchTallyWords + kNoAllocMaxChunks];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};   // device address of `scratch` on each GPU
}  // namespace

// Called with `device` current.  The array's name is only a handle on the host;
// its address in a GPU's memory comes from hipGetSymbolAddress, per GPU.
hipError_t noalloc_scratch_init(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[device]) return hipSuccess;
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = p;
    return hipSuccess;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_build_chunk_table(hipStream_t s, const void* descs, uint32_t n_obj,
                                    uint32_t* chunk_first, void* chunk_desc, uint32_t n_tiles)
{
    if (n_tiles == 0) return hipSuccess;
    const uint32_t n_chunks = n_tiles * kChunksPerTile;
    const uint32_t blocks = (n_chunks + 255u) / 256u;
    build_chunk_table_kernel<<<blocks, 256, 0, s>>>(static_cast<const u32x4*>(descs), n_obj, chunk_first,
                                                    static_cast<u32x4*>(chunk_desc), n_chunks);
    return hipGetLastError();
}

uint32_t compact_blocks(uint32_t n_obj) { return (n_obj + kRecsPerBlock - 1u) / kRecsPerBlock; }

hipError_t launch_expand_descs(hipStream_t s, const void* recs_dev, uint32_t n_obj, const void* segs_dev, uint32_t n_segs,
                               unsigned long long* block_units_dev, void* descs_dev)
{
    if (n_obj == 0 || n_segs == 0) return hipSuccess;
    const uint32_t n_blocks = compact_blocks(n_obj);
    const RecTriple* recs = static_cast<const RecTriple*>(recs_dev);
    rec_block_units_kernel<<<n_blocks, 256, 0, s>>>(recs, n_obj, block_units_dev);
    scan_block_units_kernel<<<1, 1024, 0, s>>>(block_units_dev, n_blocks);
    expand_descs_kernel<<<n_blocks, 256, 0, s>>>(recs, n_obj, static_cast<const oval_obj_segment*>(segs_dev), n_segs,
                                                 block_units_dev, static_cast<u32x4*>(descs_dev));
    return hipGetLastError();
}

uint32_t fill_tiles_per_launch(const LayoutView& lv, uint32_t range)
{
    if (range == 0 || lv.n_tiles == 0) return 1;
    const uint64_t table_bytes_per_tile =
        kChunksPerTile * (4u + (lv.chunk_desc ? 64u : 0u)) + ((uint64_t)lv.n_obj * 64u + lv.n_tiles - 1u) / lv.n_tiles;
    uint64_t max_tiles = kWarmTableBudget / table_bytes_per_tile;
    if (max_tiles > kMaxTilesPerLaunch) max_tiles = kMaxTilesPerLaunch;
    if (max_tiles < 1u) max_tiles = 1u;
    const uint32_t n_launches = (uint32_t)((range + max_tiles - 1u) / max_tiles);
    return (range + n_launches - 1u) / n_launches;
}

// A launch covers whole tiles; units of the last tile beyond the span map to
// "no more objects" in the chunk table and are neither stored nor loaded.
hipError_t launch_fill(hipStream_t s, const LaunchCfg& cfg, const LayoutView& lv, void* body,
                       uint32_t tile_begin, uint32_t tile_end)
{
    if (tile_end > lv.n_tiles) tile_end = lv.n_tiles;
    if (tile_end <= tile_begin || lv.n_obj == 0) return hipSuccess;
    sdesc_ptr sd = (sdesc_ptr)lv.descs;
    sidx_ptr cf = (sidx_ptr)lv.chunk_first;
    u32x4* b = static_cast<u32x4*>(body);
    sdesc_ptr cd = (sdesc_ptr)lv.chunk_desc;                    // NULL: two-level lookup
    // One launch holds at most 2^31 threads (HIP refuses 2^32): 32 GiB of body for
    // the fill.  It also covers no more body than its lookup tables, read by the
    // pre-pass, stay in the memory-side cache for: a 38 GB batch of 4k-16k objects
    // (0.9 GB of tables) filled at 4.65 TB/s in one launch and with 128 or 256 MB of
    // tables per launch, at 6.8 with 64 MB, at 7.1 with 32 MB, at 7.0 with 16 MB.
    // The range is cut into equal launches.
    const uint32_t per_launch = fill_tiles_per_launch(lv, tile_end - tile_begin);
    for (uint32_t t0 = tile_begin; t0 < tile_end; t0 += per_launch) {
        const uint32_t t1 = tile_end - t0 > per_launch ? t0 + per_launch : tile_end;
        const uint32_t unit_begin = t0 * kUnitsPerTile;
        const uint32_t n_units = (t1 - t0) * kUnitsPerTile;
        // (a launch of a few MiB is over before cold lookups could matter)
        if (t1 - t0 >= kWarmMinTiles)
            warm_tables_kernel<<<1024, 256, 0, s>>>(static_cast<const u32x4*>(lv.descs), lv.chunk_first, lv.n_obj,
                                                    t0 * kChunksPerTile, t1 * kChunksPerTile,
                                                    static_cast<const u32x4*>(lv.chunk_desc));
#define OVAL_LAUNCH_FILL(T)                                                                                    \
    do {                                                                                                       \
        if (lv.small_fill)                                                                                     \
            fill_small_kernel<T><<<n_units / (T / 4), T, 0, s>>>(sd, static_cast<const u32x4*>(lv.descs), cf,   \
                                                                 lv.n_obj, b, unit_begin / kUnitsPerChunk);    \
        else if (cd) fill_kernel<T, true><<<n_units / (T / 16), T, 0, s>>>(sd, cf, cd, lv.n_obj, b, unit_begin); \
        else fill_kernel<T, false><<<n_units / (T / 16), T, 0, s>>>(sd, cf, cd, lv.n_obj, b, unit_begin);       \
    } while (0)
        switch (cfg.fill_threads) {
        case 128: OVAL_LAUNCH_FILL(128); break;
        case 512: OVAL_LAUNCH_FILL(512); break;
        case 1024: OVAL_LAUNCH_FILL(1024); break;
        default: OVAL_LAUNCH_FILL(256); break;
        }
#undef OVAL_LAUNCH_FILL
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
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

hipError_t launch_gather_first_bad(hipStream_t s, const oval_obj_result* results, uint32_t n_obj, uint32_t* table)
{
    if (n_obj == 0) return hipSuccess;
    uint32_t blocks = (n_obj + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    gather_first_bad_kernel<<<blocks, 256, 0, s>>>(reinterpret_cast<const u32x4*>(results), n_obj, table);
    return hipGetLastError();
}

// `stripes`: kTallyStripes * kTallyStride zeroed u64 owned by the stream's context
// (left zeroed again); unused when the caller wants no counters.
hipError_t launch_validate(hipStream_t s, const LaunchCfg& cfg, const LayoutView& lv,
                           const void* body, uint32_t tile_begin, uint32_t tile_end,
                           oval_obj_result* results, unsigned long long* counters, unsigned long long* stripes)
{
    if (tile_end > lv.n_tiles) tile_end = lv.n_tiles;
    if (tile_end <= tile_begin || lv.n_obj == 0) return hipSuccess;
    sdesc_ptr sd = (sdesc_ptr)lv.descs;
    sidx_ptr cf = (sidx_ptr)lv.chunk_first;
    const u32x4* b = static_cast<const u32x4*>(body);
    unsigned long long* tallies = counters ? stripes : nullptr;
    for (uint32_t t0 = tile_begin; t0 < tile_end; t0 += kMaxTilesPerLaunch) {
        const uint32_t t1 = tile_end - t0 > kMaxTilesPerLaunch ? t0 + kMaxTilesPerLaunch : tile_end;
        const uint32_t chunk_begin = t0 * kChunksPerTile;
        const uint32_t n_chunks = (t1 - t0) * kChunksPerTile;
#define OVAL_LAUNCH_VAL(T)                                                                                              \
    do {                                                                                                                \
        if (lv.big_validate)                                                                                            \
            validate_big_kernel<<<n_chunks / 4u, kBigThreads, 0, s>>>(sd, cf, lv.n_obj, lv.span_units, b, chunk_begin,   \
                                                                      results, tallies);                                \
        else if (lv.small_validate)                                                                                     \
            validate_small_kernel<T><<<n_chunks / (T / 64), T, 0, s>>>(sd, static_cast<const u32x4*>(lv.descs), cf,      \
                                                                       lv.n_obj, lv.span_units, b, chunk_begin, results, \
                                                                       tallies);                                        \
        else                                                                                                            \
            validate_kernel<T><<<n_chunks / (T / 64), T, 0, s>>>(sd, cf, lv.n_obj, lv.span_units, b, chunk_begin,        \
                                                                 results, tallies);                                     \
    } while (0)
        switch (cfg.validate_threads) {
        case 128: OVAL_LAUNCH_VAL(128); break;
        case 512: OVAL_LAUNCH_VAL(512); break;
        case 1024: OVAL_LAUNCH_VAL(1024); break;
        default: OVAL_LAUNCH_VAL(256); break;
        }
#undef OVAL_LAUNCH_VAL
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (!counters) return hipSuccess;
    fold_tallies_kernel<<<1, 256, 0, s>>>(stripes, counters);
    return hipGetLastError();
}

hipError_t launch_host_batch(hipStream_t s, const void* vdescs, uint32_t n_val, const void* fdescs, uint32_t n_fill,
                             const void* up_host, void* down_host, oval_obj_result* results_host,
                             uint32_t* done_count_dev, uint32_t* done_host, uint32_t done_seq)
{
    if (n_val + n_fill == 0) return hipSuccess;
    if (n_val + n_fill == 1) {
        const u32x4* d = static_cast<const u32x4*>(n_val ? vdescs : fdescs);   // pinned host memory: read here
        host_one_kernel<<<1, 256, 0, s>>>(d[0], d[1], d[2], d[3], n_val, static_cast<const u32x4*>(up_host),
                                          static_cast<u32x4*>(down_host), reinterpret_cast<u32x4*>(results_host),
                                          done_count_dev, done_host, done_seq);
        return hipGetLastError();
    }
    if (n_val + n_fill <= kArgJobs) {
        ArgDescs a;                                             // descriptors are pinned host memory: read here
        std::memcpy(a.q, vdescs, (size_t)n_val * 64u);
        std::memcpy(a.q + 4u * n_val, fdescs, (size_t)n_fill * 64u);
        host_args_kernel<<<n_val + n_fill, 256, 0, s>>>(a, n_val, static_cast<const u32x4*>(up_host),
                                                        static_cast<u32x4*>(down_host),
                                                        reinterpret_cast<u32x4*>(results_host), done_count_dev, done_host,
                                                        done_seq);
        return hipGetLastError();
    }
    host_batch_kernel<<<n_val + n_fill, 256, 0, s>>>((sdesc_ptr)vdescs, (sdesc_ptr)fdescs, n_val,
                                                     static_cast<const u32x4*>(up_host), static_cast<u32x4*>(down_host),
                                                     reinterpret_cast<u32x4*>(results_host), done_count_dev, done_host,
                                                     done_seq);
    return hipGetLastError();
}

// oval_validate for a caller that cannot allocate: the chunk table and the
// tallies live in `scratch`, addressed through the pointer kept for `device`.
// Every call on a GPU uses that GPU's one array: the tallies are cleared and the
// chunk table rebuilt here, in stream order, and calls that overlap on a GPU race.
hipError_t launch_validate_noalloc(hipStream_t s, int device, const LaunchCfg& cfg, const void* descs_dev,
                                   uint32_t n_obj, uint32_t span_units, const void* body,
                                   oval_obj_result* results, unsigned long long* counters)
{
    void* base = nullptr;
    if (device >= 0 && device < kMaxDevices) {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        base = g_scratch_ptr[device];
    }
    if (!base) return hipErrorNotInitialized;
    unsigned long long* const tallies = static_cast<unsigned long long*>(base);
    uint32_t* const chunk_first = static_cast<uint32_t*>(base) + kScratchTallyWords;
    LayoutView lv;
    lv.descs = descs_dev;
    lv.chunk_desc = nullptr;
    lv.chunk_first = chunk_first;
    lv.n_obj = n_obj;
    lv.n_tiles = (span_units + kUnitsPerTile - 1u) / kUnitsPerTile;
    lv.span_units = span_units;
    lv.total_units = 0;
    lv.small_fill = false;
    lv.small_validate = false;
    lv.big_validate = false;
    if (lv.n_tiles > kNoAllocMaxTiles) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(tallies, 0, kTallyBytes, s);
    if (e != hipSuccess) return e;
    if (n_obj) {
        e = launch_build_chunk_table(s, descs_dev, n_obj, chunk_first, nullptr, lv.n_tiles);
        if (e != hipSuccess) return e;
    }
    uint32_t blocks = (n_obj + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0u) blocks = 1u;
    init_results_sum_kernel<<<blocks, 256, 0, s>>>(reinterpret_cast<u32x4*>(results),
                                                   static_cast<const u32x4*>(descs_dev), n_obj, counters);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_validate(s, cfg, lv, body, 0, lv.n_tiles, results, counters, tallies);
}

}  // namespace oval
