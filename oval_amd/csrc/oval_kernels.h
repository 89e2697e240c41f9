// oval_kernels.h — internal launch interface between the C ABI (oval_capi.cpp)
// and the gfx950 kernels (oval_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/oval_hip.h"

namespace oval {

// One wavefront instruction covers 4 consecutive units (1 KiB); a tile is the
// granularity of the tile table and of host-staging slices.
constexpr uint32_t kUnitsPerTile  = 256;                   // 64 KiB
constexpr uint32_t kBlockThreads  = 256;                   // 4 wavefronts

struct LayoutView {
    const void*     descs;        // oval_obj_desc[n_obj] in device memory
    const uint32_t* tile_first;   // [n_tiles] first object whose end lies beyond the tile's first unit
    uint32_t        n_obj;
    uint32_t        n_tiles;
    uint32_t        span_units;   // units from offset 0 to the end of the last object
    uint64_t        total_units;  // sum of n_units
};

struct LaunchCfg {
    int grid_blocks;   // number of 256-thread workgroups
    int variant;       // bit 0: XCD-contiguous tile order; bits 4-5: unroll / tiles per pass (benchmarks)
};

hipError_t launch_build_tile_table(hipStream_t s, const void* descs, uint32_t n_obj,
                                   uint32_t* tile_first, uint32_t n_tiles);
hipError_t launch_fill(hipStream_t s, const LaunchCfg& cfg, const LayoutView& lv, void* body,
                       uint32_t tile_begin, uint32_t tile_end);
hipError_t launch_init_results(hipStream_t s, const LayoutView& lv, oval_obj_result* results,
                               unsigned long long* counters);
hipError_t launch_validate(hipStream_t s, const LaunchCfg& cfg, const LayoutView& lv,
                           const void* body, uint32_t tile_begin, uint32_t tile_end,
                           oval_obj_result* results, unsigned long long* counters);

}  // namespace oval
