// scalar_fetch_probe.hip — what FETCH_SIZE reports for the table lookups of the
// fill kernel: 64-byte scalar loads (s_load_dwordx16), one record per 4 KiB chunk,
// neighbouring records read by neighbouring workgroups.  Two questions:
//   1. are 64-byte scalar misses tallied like the 128-byte requests of a wide
//      streaming read (FETCH_SIZE x 2 = bytes) or at their own size?
//   2. do two workgroups that read the two halves of one 128-byte line cost one
//      fetch when they run on the same XCD (workgroups b and b + 8)?
// Every kernel touches each 64-byte record of a table larger than the memory-side
// cache exactly once; run each under `rocprofv3 --pmc FETCH_SIZE` (tools/scalar_fetch_probe.sh)
// and compare KiB reported with the table size.  Not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/scalar_fetch_probe tools/scalar_fetch_probe.hip
//   build/scalar_fetch_probe <variant> [table MiB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4* sptr;

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

__device__ __forceinline__ uint32_t rec_sum(sptr t, uint32_t rec)
{
    const u32x4 a = t[4u * rec], b = t[4u * rec + 1u], c = t[4u * rec + 2u], d = t[4u * rec + 3u];
    return a.x ^ b.y ^ c.z ^ d.w;
}

// one record per workgroup, in workgroup order (what the fill does)
__global__ __launch_bounds__(256) void scalar_one(sptr t, uint32_t* sink)
{
    const uint32_t v = rec_sum(t, blockIdx.x);
    if (v == 0x12345u) sink[0] = v;
}

// the same with workgroups b and b + 8 of every sixteen on neighbouring records
__global__ __launch_bounds__(256) void scalar_pair(sptr t, uint32_t* sink)
{
    uint32_t b = blockIdx.x;
    b = (b & ~15u) | ((b & 7u) << 1) | ((b >> 3) & 1u);
    const uint32_t v = rec_sum(t, b);
    if (v == 0x12345u) sink[0] = v;
}

// two neighbouring records (one 128-byte line) per workgroup: certainly one XCD
__global__ __launch_bounds__(256) void scalar_two(sptr t, uint32_t* sink)
{
    const uint32_t v = rec_sum(t, 2u * blockIdx.x) ^ rec_sum(t, 2u * blockIdx.x + 1u);
    if (v == 0x12345u) sink[0] = v;
}

// (the record kernels are launched with one wavefront per workgroup: 2^24 workgroups of 256 threads would exceed a grid)
// calibration: the same bytes with 16-byte-per-lane vector loads, 4 KiB per workgroup
__global__ __launch_bounds__(256) void wide(const u32x4* t, uint32_t* sink)
{
    const u32x4 v = t[(size_t)blockIdx.x * 256u + threadIdx.x];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) sink[0] = v.x;
}

// one 64-byte record per workgroup with vector loads (4 lanes x 16 bytes)
__global__ __launch_bounds__(256) void vector_one(const u32x4* t, uint32_t* sink)
{
    if (threadIdx.x < 4) {
        const u32x4 v = t[(size_t)blockIdx.x * 4u + threadIdx.x];
        if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) sink[0] = v.x;
    }
}

int main(int argc, char** argv)
{
    const char* variant = argc > 1 ? argv[1] : "scalar_one";
    const size_t mib = argc > 2 ? (size_t)atoll(argv[2]) : 1024;
    const size_t bytes = mib << 20;
    void* table = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMalloc((void**)&sink, 64));
    CHECK(hipMemset(table, 0x5A, bytes));
    CHECK(hipDeviceSynchronize());
    const uint32_t recs = (uint32_t)(bytes / 64);
    if (!strcmp(variant, "scalar_one")) scalar_one<<<recs, 64>>>((sptr)table, sink);
    else if (!strcmp(variant, "scalar_pair")) scalar_pair<<<recs, 64>>>((sptr)table, sink);
    else if (!strcmp(variant, "scalar_two")) scalar_two<<<recs / 2, 64>>>((sptr)table, sink);
    else if (!strcmp(variant, "vector_one")) vector_one<<<recs, 64>>>((const u32x4*)table, sink);
    else if (!strcmp(variant, "wide")) wide<<<(uint32_t)(bytes / 4096), 256>>>((const u32x4*)table, sink);
    else { fprintf(stderr, "unknown variant %s\n", variant); return 2; }
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    printf("%s: table %zu MiB = %zu KiB, %u records of 64 bytes\n", variant, mib, bytes >> 10, recs);
    return 0;
}
