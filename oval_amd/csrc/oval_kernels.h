// oval_kernels.h — internal launch interface between the C ABI (oval_capi.cpp)
// and the gfx950 kernels (oval_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/oval_hip.h"

namespace oval {

// One wavefront instruction covers 4 consecutive units (1 KiB).  A chunk is the
// granularity of the object lookup table and what one fill workgroup writes; a
// tile is what a validating wavefront takes per pass and the unit of
// host-staging slices.
constexpr uint32_t kUnitsPerChunk = 16;                    // 4 KiB
constexpr uint32_t kChunksPerTile = 16;
constexpr uint32_t kUnitsPerTile  = kUnitsPerChunk * kChunksPerTile;   // 64 KiB
constexpr uint32_t kMaxTilesPerLaunch = 1u << 19;          // 32 GiB of body per kernel launch
constexpr uint32_t kWarmMinTiles = 64;                    // fills under 4 MiB skip the table pre-pass
constexpr uint64_t kWarmTableBudget = 32ull << 20;         // lookup-table bytes one fill launch may depend on

// The entry point that allocates nothing keeps its chunk table in a static
// device array sized for this much body: 65,536 tiles = 4 GiB.
constexpr uint32_t kNoAllocMaxTiles = 1u << 16;

// Bad-unit tallies are spread over this many copies of the counter vector, one
// 128-byte line each, and folded into the caller's counters after the pass.
constexpr uint32_t kTallyStripes = 256;
constexpr uint32_t kTallyStride  = 16;                     // u64 per stripe
constexpr size_t   kTallyBytes   = (size_t)kTallyStripes * kTallyStride * 8;

struct LayoutView {
    const void*     descs;        // oval_obj_desc[n_obj] in device memory
    const void*     chunk_desc;   // [n_tiles * kChunksPerTile] 64-byte copy of that object's descriptor, or NULL
    const uint32_t* chunk_first;  // [n_tiles * kChunksPerTile] first object whose end lies beyond the chunk's first unit
    uint32_t        n_obj;
    uint32_t        n_tiles;
    uint32_t        span_units;   // units from offset 0 to the end of the last object
    uint64_t        total_units;  // sum of n_units
    // objects of a few units: the kernels that look all objects of a chunk up at once (chosen per pass:
    // the fill gains from them up to larger objects than the validate pass does)
    bool            small_fill, small_validate;
    // objects far larger than a workgroup's range: the validate kernel that reads with one load per thread from
    // workgroups of 1024 threads and looks up once per workgroup
    bool            big_validate;
};

struct LaunchCfg {
    int fill_threads;       // workgroup size of the fill kernel: 128, 256 (shipped), 512, 1024
    int validate_threads;   // workgroup size of the validate kernel, likewise
};

hipError_t launch_build_chunk_table(hipStream_t s, const void* descs, uint32_t n_obj,
                                    uint32_t* chunk_first, void* chunk_desc, uint32_t n_tiles);
// The descriptor table written on the device from oval_obj_rec[n_obj] and
// oval_obj_segment[n_segs] (both in device memory; the segments cover the
// records in order): bodies back to back from offset 0.  `block_units_dev` is
// scratch of compact_blocks(n_obj) u64.
uint32_t compact_blocks(uint32_t n_obj);
hipError_t launch_expand_descs(hipStream_t s, const void* recs_dev, uint32_t n_obj, const void* segs_dev, uint32_t n_segs,
                               unsigned long long* block_units_dev, void* descs_dev);
// Tiles one fill launch covers when `range` tiles are to be filled: bounded by
// the HIP grid limit and by the lookup-table bytes a launch may depend on, and
// chosen so that the launches of the range are of equal size.
uint32_t fill_tiles_per_launch(const LayoutView& lv, uint32_t range);
hipError_t launch_fill(hipStream_t s, const LaunchCfg& cfg, const LayoutView& lv, void* body,
                       uint32_t tile_begin, uint32_t tile_end);
hipError_t launch_init_results(hipStream_t s, const LayoutView& lv, oval_obj_result* results,
                               unsigned long long* counters);
// table[i] = results[i].first_bad_unit for i < n_obj
hipError_t launch_gather_first_bad(hipStream_t s, const oval_obj_result* results, uint32_t n_obj, uint32_t* table);
hipError_t launch_validate(hipStream_t s, const LaunchCfg& cfg, const LayoutView& lv,
                           const void* body, uint32_t tile_begin, uint32_t tile_end,
                           oval_obj_result* results, unsigned long long* counters,
                           unsigned long long* stripes);

// One launch for a small batch whose descriptors, bodies and results all live in
// pinned host memory: n_val validate jobs (bodies at byte offsets into up_host,
// one result each into results_host) and n_fill fill jobs (into down_host).
// Descriptor offsets and lengths are the caller's responsibility: nothing is checked.
// With `done_host` (pinned) the last workgroup to finish stores `done_seq` there
// after everything the batch wrote is visible to the host; `done_count_dev` is a
// zeroed u32 in device memory that the launch leaves zeroed.  NULL: no such signal.
hipError_t launch_host_batch(hipStream_t s, const void* vdescs, uint32_t n_val, const void* fdescs, uint32_t n_fill,
                             const void* up_host, void* down_host, oval_obj_result* results_host,
                             uint32_t* done_count_dev, uint32_t* done_host, uint32_t done_seq);

// Looks up, once per GPU and with that GPU current, where the static scratch
// array of launch_validate_noalloc lives in its memory.
hipError_t noalloc_scratch_init(int device);
hipError_t launch_validate_noalloc(hipStream_t s, int device, const LaunchCfg& cfg, const void* descs_dev,
                                   uint32_t n_obj, uint32_t span_units, const void* body,
                                   oval_obj_result* results, unsigned long long* counters);

}  // namespace oval
