// write_shapes.hip — second write-stream probe: how the runtime's own fill
// kernel (hipMemsetAsync) is launched, and whether short-lived workgroups that
// sweep the buffer in dispatch order beat persistent wavefronts that each own a
// 64 KiB tile.  Run under `rocprofv3 --kernel-trace` to see the runtime kernel's
// grid and workgroup size.  Not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -o write_shapes tools/write_shapes.hip && ./write_shapes [GiB]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

template <bool NTS>
__device__ __forceinline__ void st(u32x4* p, u32x4 v)
{
    if (NTS) __builtin_nontemporal_store(v, p); else *p = v;
}

// One workgroup of THREADS threads writes KIB contiguous KiB and exits; the
// grid covers the buffer.  Instruction k of the workgroup covers THREADS*16 B.
template <int THREADS, int KIB, bool NTS>
__global__ __launch_bounds__(THREADS) void once_kernel(u32x4* __restrict__ buf)
{
    constexpr int PER = KIB * 64 / THREADS;                    // stores per thread
    u32x4* p = buf + (size_t)blockIdx.x * (KIB * 64) + threadIdx.x;
    u32x4 v = {(uint32_t)threadIdx.x, 1u, 2u, 3u};
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        v.x += 256u;
        st<NTS>(p + (size_t)k * THREADS, v);
    }
}

// The same, but every wavefront of the workgroup owns a contiguous KIB/waves
// stretch (wave-contiguous instead of workgroup-interleaved).
template <int THREADS, int KIB, bool NTS>
__global__ __launch_bounds__(THREADS) void once_wave_kernel(u32x4* __restrict__ buf)
{
    constexpr int WAVES = THREADS / 64;
    constexpr int PER = KIB / WAVES;                            // KiB (= stores) per wave
    u32x4* p = buf + (size_t)blockIdx.x * (KIB * 64) + (size_t)(threadIdx.x >> 6) * (PER * 64) + (threadIdx.x & 63);
    u32x4 v = {(uint32_t)threadIdx.x, 1u, 2u, 3u};
#pragma unroll 16
    for (int k = 0; k < PER; ++k) {
        v.x += 256u;
        st<NTS>(p + (size_t)k * 64, v);
    }
}

// once_kernel <256 thr, 4 KiB> with (a) a dependent scalar load chain in front of
// the store, as the pattern kernels have (table -> 64-byte entry -> store), and
// (b) dynamic LDS to bound the workgroups resident per CU.
typedef const __attribute__((address_space(4))) u32x4* sc_u32x4;
typedef const __attribute__((address_space(4))) uint32_t* sc_u32;
// CHAIN 1: index per 64 KiB -> entry (mostly cache hits).  CHAIN 2: index per 4 KiB chunk -> an entry
// every ~2.3 chunks (what a batch of 4-16 KiB objects looks like: both loads often miss the XCD's L2).
// CHAIN 3: no index, one 64-byte entry per chunk read directly (one load, always a miss).
template <int CHAIN>
__global__ __launch_bounds__(256) void once_chain_kernel(u32x4* __restrict__ buf, sc_u32 first, sc_u32x4 table)
{
    extern __shared__ uint32_t dyn_lds[];
    u32x4 v = {(uint32_t)threadIdx.x, 1u, 2u, 3u};
    if (CHAIN == 1) {
        const uint32_t e = first[blockIdx.x >> 4];
        const u32x4 d = table[4u * e];
        v.y += d.x;
    } else if (CHAIN == 2) {
        const uint32_t e = first[blockIdx.x];
        const u32x4 d = table[4u * e];
        v.y += d.x;
    } else if (CHAIN == 3) {
        const u32x4 d = table[4u * blockIdx.x];
        v.y += d.x;
    }
    if (buf == nullptr) dyn_lds[threadIdx.x] = v.x;              // keeps the LDS allocation alive; never taken
    __builtin_nontemporal_store(v, buf + (size_t)blockIdx.x * 256 + threadIdx.x);
}

// Read counterpart of once_kernel: PER loads per thread, all issued before use.
template <int THREADS, int KIB, bool NTL>
__global__ __launch_bounds__(THREADS) void once_read_kernel(const u32x4* __restrict__ buf, uint32_t* __restrict__ sink)
{
    constexpr int PER = KIB * 64 / THREADS;
    const u32x4* p = buf + (size_t)blockIdx.x * (KIB * 64) + threadIdx.x;
    u32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = NTL ? __builtin_nontemporal_load(p + (size_t)k * THREADS) : p[(size_t)k * THREADS];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) acc |= (v[k].x ^ v[k].y) | (v[k].z ^ v[k].w);
    if (acc == 0x12345678u) sink[0] = acc;
}

// wave-contiguous variant: each wavefront reads KIB/waves consecutive KiB
template <int THREADS, int KIB, bool NTL>
__global__ __launch_bounds__(THREADS) void once_read_wave_kernel(const u32x4* __restrict__ buf, uint32_t* __restrict__ sink)
{
    constexpr int PER = KIB / (THREADS / 64);
    const u32x4* p = buf + (size_t)blockIdx.x * (KIB * 64) + (size_t)(threadIdx.x >> 6) * (PER * 64) + (threadIdx.x & 63);
    u32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = NTL ? __builtin_nontemporal_load(p + (size_t)k * 64) : p[(size_t)k * 64];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) acc |= (v[k].x ^ v[k].y) | (v[k].z ^ v[k].w);
    if (acc == 0x12345678u) sink[0] = acc;
}

// Persistent wavefronts, 64 KiB tile each per pass (the library's shape), with
// optional dependent scalar load chain at the start of every tile (table ->
// entry -> first store), as the pattern kernels have.
typedef const __attribute__((address_space(4))) u32x4* stable_ptr;
template <bool CHAIN, bool NTS>
__global__ __launch_bounds__(256) void persistent_kernel(u32x4* __restrict__ buf, uint32_t n_tiles,
                                                         const uint32_t* __restrict__ first, const u32x4* __restrict__ table)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    stable_ptr stab = (stable_ptr)table;
    u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    for (uint32_t t = wave; t < n_tiles; t += n_waves) {
        if (CHAIN) {
            const uint32_t e = __builtin_amdgcn_readfirstlane(first[t]);
            const u32x4 d = stab[4u * e];
            v.y += d.x;
        }
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {
            v.x += 256u;
            st<NTS>(&buf[((size_t)t * 64 + k) * 64 + lane], v);
        }
    }
}

template <class F>
void time_it(const char* name, size_t bytes, F launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int r = 0; r < 7; ++r) {
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float m;
        CHECK(hipEventElapsedTime(&m, a, b));
        if (r >= 2) ms.push_back(m);
    }
    CHECK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    printf("%-52s best %8.1f GB/s   median %8.1f GB/s\n", name, bytes / (ms.front() * 1e-3) / 1e9,
           bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
    fflush(stdout);
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
}

int main(int argc, char** argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 8;
    const size_t bytes = gib << 30;
    const uint32_t kib = (uint32_t)(bytes >> 10);
    const uint32_t n_tiles = kib / 64;
    u32x4* buf;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 1, bytes));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, buffer %zu GiB\n", prop.gcnArchName, cus, gib);

    time_it("hipMemsetAsync (byte value)", bytes, [&] { CHECK(hipMemsetAsync(buf, 0x5A, bytes)); });
    time_it("hipMemsetD32Async", bytes, [&] { CHECK(hipMemsetD32Async((hipDeviceptr_t)buf, 0x5A5A5A5A, bytes / 4)); });

#define ONCE(T, K, N, LABEL)                                                                    \
    time_it("once  wg-interleaved " #T " thr x " #K " KiB " LABEL, bytes,                        \
            [&] { once_kernel<T, K, N><<<kib / K, T>>>(buf); });
#define ONCEW(T, K, N, LABEL)                                                                   \
    time_it("once  wave-contig    " #T " thr x " #K " KiB " LABEL, bytes,                        \
            [&] { once_wave_kernel<T, K, N><<<kib / K, T>>>(buf); });
    ONCE(256, 4, false, "plain") ONCE(256, 4, true, "nt")
    ONCE(256, 16, false, "plain") ONCE(256, 16, true, "nt")
    ONCE(256, 64, false, "plain") ONCE(256, 64, true, "nt")
    ONCE(256, 256, false, "plain") ONCE(256, 256, true, "nt")
    ONCE(1024, 16, false, "plain") ONCE(1024, 16, true, "nt")
    ONCE(1024, 64, false, "plain") ONCE(1024, 64, true, "nt")
    ONCE(1024, 256, false, "plain") ONCE(1024, 256, true, "nt")
    ONCE(64, 16, true, "nt") ONCE(64, 64, true, "nt")
    ONCE(64, 1, true, "nt") ONCE(128, 2, true, "nt") ONCE(512, 8, true, "nt") ONCE(256, 8, true, "nt") ONCE(512, 16, true, "nt")
    ONCE(64, 1, false, "plain") ONCE(128, 2, false, "plain") ONCE(512, 8, false, "plain")
    uint32_t* sink;
    CHECK(hipMalloc(&sink, 256));
#define ONCER(T, K, N, LABEL)                                                                   \
    time_it("once-read wg-interleaved " #T " thr x " #K " KiB " LABEL, bytes,                    \
            [&] { once_read_kernel<T, K, N><<<kib / K, T>>>(buf, sink); });
#define ONCERW(T, K, N, LABEL)                                                                  \
    time_it("once-read wave-contig    " #T " thr x " #K " KiB " LABEL, bytes,                    \
            [&] { once_read_wave_kernel<T, K, N><<<kib / K, T>>>(buf, sink); });
    ONCER(256, 4, true, "nt") ONCER(256, 4, false, "plain")
    ONCER(256, 8, true, "nt") ONCER(256, 16, true, "nt") ONCER(256, 16, false, "plain")
    ONCER(256, 32, true, "nt") ONCER(256, 64, true, "nt")
    ONCER(512, 8, true, "nt") ONCER(512, 16, true, "nt") ONCER(1024, 16, true, "nt") ONCER(1024, 64, true, "nt")
    ONCER(64, 1, true, "nt") ONCER(64, 4, true, "nt")
    ONCERW(256, 16, true, "nt") ONCERW(256, 32, true, "nt") ONCERW(256, 64, true, "nt")
    ONCEW(256, 64, false, "plain") ONCEW(256, 64, true, "nt")
    ONCEW(256, 256, false, "plain") ONCEW(256, 256, true, "nt")
    ONCEW(1024, 256, true, "nt") ONCEW(1024, 1024, true, "nt")

    {
        const uint32_t n_chunks = kib / 4;
        std::vector<uint32_t> first(n_chunks);
        for (uint32_t c = 0; c < n_chunks; ++c) first[c] = (uint32_t)(((uint64_t)c * 10) / 23);
        const size_t n_entries = n_chunks;
        uint32_t* d_first;
        u32x4* table;
        CHECK(hipMalloc(&d_first, n_chunks * 4));
        CHECK(hipMalloc(&table, n_entries * 64));
        CHECK(hipMemcpy(d_first, first.data(), n_chunks * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(table, 0, n_entries * 64));
        for (int lds_kib : {0, 40}) {
            char name[128];
#define OC(C, LABEL)                                                                             \
            snprintf(name, sizeof name, "once 256 thr x 4 KiB nt, %-28s %2d KiB LDS/wg", LABEL, lds_kib); \
            time_it(name, bytes, [&] { once_chain_kernel<C><<<kib / 4, 256, lds_kib * 1024>>>(buf, (sc_u32)d_first, (sc_u32x4)table); });
            OC(0, "no chain") OC(1, "2 loads, mostly hits") OC(2, "2 loads, per-chunk index") OC(3, "1 load, 64 B per chunk")
#undef OC
        }
        CHECK(hipFree(d_first));
        CHECK(hipFree(table));
    }
    {
        // tile -> entry table like the library's: ascending, ~5.4 entries per tile, 64-byte entries
        std::vector<uint32_t> first(n_tiles);
        for (uint32_t t = 0; t < n_tiles; ++t) first[t] = (uint32_t)(((uint64_t)t * 27) / 4);
        const size_t n_entries = (size_t)first.back() + 2;
        uint32_t* d_first;
        u32x4* table;
        CHECK(hipMalloc(&d_first, n_tiles * 4));
        CHECK(hipMalloc(&table, n_entries * 64));
        CHECK(hipMemcpy(d_first, first.data(), n_tiles * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(table, 0, n_entries * 64));
        for (int bpc : {4, 8}) {
            char name[128];
            const int grid = cus * bpc;
            snprintf(name, sizeof name, "persistent 64 KiB tiles nt, no chain      %d b/CU", bpc);
            time_it(name, bytes, [&] { persistent_kernel<false, true><<<grid, 256>>>(buf, n_tiles, d_first, table); });
            snprintf(name, sizeof name, "persistent 64 KiB tiles nt, tile chain    %d b/CU", bpc);
            time_it(name, bytes, [&] { persistent_kernel<true, true><<<grid, 256>>>(buf, n_tiles, d_first, table); });
            snprintf(name, sizeof name, "persistent 64 KiB tiles plain, no chain   %d b/CU", bpc);
            time_it(name, bytes, [&] { persistent_kernel<false, false><<<grid, 256>>>(buf, n_tiles, d_first, table); });
            snprintf(name, sizeof name, "persistent 64 KiB tiles plain, tile chain %d b/CU", bpc);
            time_it(name, bytes, [&] { persistent_kernel<true, false><<<grid, 256>>>(buf, n_tiles, d_first, table); });
        }
        CHECK(hipFree(d_first));
        CHECK(hipFree(table));
    }
    CHECK(hipFree(buf));
    return 0;
}
