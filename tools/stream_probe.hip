// stream_probe.hip — measures what MI355X sustains for pure write-only and
// read-only 16-byte-per-lane streams, in the access shapes the pattern kernels
// can choose between.  Not part of the library; used to pick the shapes and to
// state the practical ceiling next to the 8 TB/s datasheet peak.
//
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe tools/stream_probe.hip && ./stream_probe [GiB]
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

enum Flavor { PLAIN = 0, NT = 1, SC1 = 2, SC0SC1 = 3, NTSC1 = 4 };
enum Shape { CONTIG = 0, SEG4 = 1 };

// A wave-tile is 64 KiB.  CONTIG: instruction k of a wave covers the 1 KiB at
// tile + k KiB.  SEG4: lane group g (16 lanes) walks the 16 KiB at tile + g*16 KiB,
// 256 B per instruction.
template <int SHAPE>
__device__ __forceinline__ size_t lane_index(size_t tile, int k, int lane)
{
    if (SHAPE == CONTIG) return tile * 4096 + (size_t)k * 64 + lane;            // in 16-B elements
    return tile * 4096 + (size_t)(lane >> 4) * 1024 + (size_t)k * 16 + (lane & 15);
}

template <int FLAVOR>
__device__ __forceinline__ void store16(u32x4* p, u32x4 v)
{
    if (FLAVOR == PLAIN) *p = v;
    else if (FLAVOR == NT) __builtin_nontemporal_store(v, p);
    else if (FLAVOR == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if (FLAVOR == SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int SHAPE, int FLAVOR>
__global__ __launch_bounds__(256) void write_kernel(u32x4* __restrict__ buf, uint32_t n_tiles)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    for (uint32_t t = wave; t < n_tiles; t += n_waves) {
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {
            v.x += 256u;
            store16<FLAVOR>(&buf[lane_index<SHAPE>(t, k, lane)], v);
        }
    }
}

template <int SHAPE, bool NTLOAD, int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ buf, uint32_t n_tiles,
                                                   uint32_t* __restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    uint32_t acc = 0;
    for (uint32_t t = wave; t < n_tiles; t += n_waves) {
        for (int kb = 0; kb < 64; kb += UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
                const u32x4* p = &buf[lane_index<SHAPE>(t, kb + i, lane)];
                v[i] = NTLOAD ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) acc |= (v[i].x ^ v[i].y) | (v[i].z ^ v[i].w);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;   // never true for the data written; keeps the loads alive
}

// Same streams with a configurable tile: a wave writes/reads TK consecutive KiB,
// then jumps by (number of waves) tiles.  TK = 1 is a plain grid-stride sweep.
template <int TK, bool NTS>
__global__ __launch_bounds__(256) void write_tile_kernel(u32x4* __restrict__ buf, uint32_t n_tiles)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    for (uint32_t t = wave; t < n_tiles; t += n_waves) {
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            v.x += 256u;
            u32x4* p = &buf[((size_t)t * TK + k) * 64 + lane];
            if (NTS) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

template <int TK, int UNROLL>
__global__ __launch_bounds__(256) void read_tile_kernel(const u32x4* __restrict__ buf, uint32_t n_tiles,
                                                        uint32_t* __restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    uint32_t acc = 0;
    for (uint32_t t = wave; t < n_tiles; t += n_waves) {
        for (int kb = 0; kb < TK; kb += UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int i = 0; i < UNROLL; ++i)
                v[i] = __builtin_nontemporal_load(&buf[((size_t)t * TK + kb + i) * 64 + lane]);
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) acc |= (v[i].x ^ v[i].y) | (v[i].z ^ v[i].w);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// 64 KiB write tiles with a trickle of reads mixed in: one 64-byte table entry
// per `EVERY` KiB written (EVERY = 0: none), fetched on the scalar unit one step
// ahead of use, or with a vector load (VEC) whose wait also drains the stores.
typedef const __attribute__((address_space(4))) u32x4* stable_ptr;
template <int EVERY, bool VEC>
__global__ __launch_bounds__(256) void write_mixed_kernel(u32x4* __restrict__ buf, uint32_t n_tiles,
                                                          const u32x4* __restrict__ table, uint32_t n_entries)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    stable_ptr st = (stable_ptr)table;
    u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    for (uint32_t t = wave; t < n_tiles; t += n_waves) {
        uint32_t e = (t * (64 / (EVERY ? EVERY : 64))) % n_entries;
        u32x4 cur = {0, 0, 0, 0};
        if (EVERY) cur = VEC ? table[4u * e] : st[4u * e];
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {
            if (EVERY && (k % EVERY) == EVERY - 1) {
                v.y += cur.x;                                  // consume the entry fetched earlier
                e = (e + 1u) % n_entries;
                if (VEC) {
                    cur = table[4u * e + (uint32_t)(lane & 3)];
                } else {
                    __builtin_amdgcn_sched_barrier(0);
                    cur = st[4u * e];
                }
            }
            v.x += 256u;
            buf[((size_t)t * 64 + k) * 64 + lane] = v;
        }
    }
}

// The same read volume (64 B per 8 KiB written) fetched in bursts: a wave reads
// LOADS KiB of contiguous table with vector loads, then writes LOADS*2 tiles.
template <int LOADS>
__global__ __launch_bounds__(256) void write_burst_kernel(u32x4* __restrict__ buf, uint32_t n_tiles,
                                                          const u32x4* __restrict__ table, uint32_t table_kib)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    uint32_t round = 0;
    for (uint32_t t0 = wave * (LOADS * 2); t0 < n_tiles; t0 += n_waves * (LOADS * 2), ++round) {
        uint32_t acc = 0;
        const uint32_t base = ((wave + round * n_waves) * LOADS) % (table_kib - LOADS);
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            const u32x4 d = __builtin_nontemporal_load(&table[(size_t)(base + l) * 64 + lane]);
            acc += d.x;
        }
        v.y += acc;
        for (uint32_t t = t0; t < t0 + LOADS * 2 && t < n_tiles; ++t) {
#pragma unroll 8
            for (int k = 0; k < 64; ++k) {
                v.x += 256u;
                buf[((size_t)t * 64 + k) * 64 + lane] = v;
            }
        }
    }
}

// Persistent workgroups of 1024 threads, one per CU: each owns a contiguous
// share of the buffer and alternates between loading a table slice into LDS in
// one burst (BURST_KIB) and writing the matching stretch of the buffer with no
// reads at all.  The read volume per byte written is the same as above.
template <int BURST_KIB>
__global__ __launch_bounds__(1024) void write_phased_kernel(u32x4* __restrict__ buf, uint32_t n_tiles,
                                                            const u32x4* __restrict__ table, uint32_t table_kib)
{
    __shared__ u32x4 lds[BURST_KIB * 64];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;                      // 16 waves
    const uint32_t tiles_per_block = (n_tiles + gridDim.x - 1) / gridDim.x;
    const uint32_t t_begin = blockIdx.x * tiles_per_block;
    const uint32_t t_end = min(n_tiles, t_begin + tiles_per_block);
    const uint32_t tiles_per_round = BURST_KIB * 2;              // 64 B of table per 8 KiB written
    u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    uint32_t round = 0;
    for (uint32_t r0 = t_begin; r0 < t_end; r0 += tiles_per_round, ++round) {
        const uint32_t base = ((blockIdx.x + round * gridDim.x) * BURST_KIB) % (table_kib - BURST_KIB);
        for (uint32_t i = threadIdx.x; i < BURST_KIB * 64; i += 1024)
            lds[i] = __builtin_nontemporal_load(&table[(size_t)base * 64 + i]);
        __syncthreads();
        v.y += lds[(lane * 17 + wave) % (BURST_KIB * 64)].x;
        const uint32_t r1 = min(t_end, r0 + tiles_per_round);
        for (uint32_t t = r0 + wave; t < r1; t += 16) {
#pragma unroll 8
            for (int k = 0; k < 64; ++k) {
                v.x += 256u;
                buf[((size_t)t * 64 + k) * 64 + lane] = v;
            }
        }
        __syncthreads();
    }
}

struct Result { const char* name; double best_gbs, med_gbs; };

template <class F>
Result time_it(const char* name, size_t bytes, F launch)
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
    Result res{name, bytes / (ms.front() * 1e-3) / 1e9, bytes / (ms[ms.size() / 2] * 1e-3) / 1e9};
    printf("%-44s best %8.1f GB/s   median %8.1f GB/s\n", name, res.best_gbs, res.med_gbs);
    fflush(stdout);
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return res;
}

int main(int argc, char** argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 8;
    const size_t bytes = gib << 30;
    const uint32_t n_tiles = (uint32_t)(bytes >> 16);
    u32x4 *buf, *buf2;
    uint32_t* sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&buf2, bytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipMemset(buf2, 1, bytes));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, buffer %zu GiB\n", prop.gcnArchName, cus, gib);

    time_it("hipMemsetAsync (runtime fill)", bytes, [&] { CHECK(hipMemsetAsync(buf, 0x5A, bytes)); });
    time_it("hipMemcpyDtoD (read+write bytes)", 2 * bytes,
            [&] { CHECK(hipMemcpyAsync(buf2, buf, bytes, hipMemcpyDeviceToDevice)); });

    char name[128];
    for (int bpc : {2, 4, 8, 16}) {
        const int grid = cus * bpc;
#define W(SH, FL, LABEL)                                                                         \
        snprintf(name, sizeof name, "write %-6s %-7s %2d blocks/CU", #SH, LABEL, bpc);            \
        time_it(name, bytes, [&] { write_kernel<SH, FL><<<grid, 256>>>(buf, n_tiles); });
        W(CONTIG, PLAIN, "plain") W(CONTIG, NT, "nt") W(CONTIG, SC1, "sc1") W(CONTIG, SC0SC1, "sc0sc1") W(CONTIG, NTSC1, "sc1nt")
        W(SEG4, PLAIN, "plain") W(SEG4, NT, "nt") W(SEG4, SC1, "sc1") W(SEG4, NTSC1, "sc1nt")
#undef W
    }
    for (int bpc : {2, 4, 8, 16}) {
        const int grid = cus * bpc;
#define R(SH, NTL, UN, LABEL)                                                                    \
        snprintf(name, sizeof name, "read  %-6s %-7s unroll %d %2d blocks/CU", #SH, LABEL, UN, bpc); \
        time_it(name, bytes, [&] { read_kernel<SH, NTL, UN><<<grid, 256>>>(buf, n_tiles, sink); });
        R(CONTIG, false, 4, "plain") R(CONTIG, true, 4, "nt") R(CONTIG, true, 8, "nt") R(CONTIG, true, 16, "nt")
        R(SEG4, false, 4, "plain") R(SEG4, true, 4, "nt") R(SEG4, true, 8, "nt") R(SEG4, true, 16, "nt")
#undef R
    }
    {
        const uint32_t n_entries = 1u << 20;                   // 64 MiB table of 64-byte entries
        u32x4* table;
        CHECK(hipMalloc(&table, (size_t)n_entries * 64));
        CHECK(hipMemset(table, 0, (size_t)n_entries * 64));
        for (int bpc : {2, 4, 8}) {
            const int grid = cus * bpc;
#define WM(EV, VEC, LABEL)                                                                        \
            snprintf(name, sizeof name, "write 64K + 64B read per %2d KiB %-6s %2d b/CU", EV, LABEL, bpc); \
            time_it(name, bytes, [&] { write_mixed_kernel<EV, VEC><<<grid, 256>>>(buf, n_tiles, table, n_entries); });
            WM(0, false, "none") WM(8, false, "scalar") WM(8, true, "vector") WM(4, false, "scalar") WM(16, false, "scalar") WM(16, true, "vector")
#undef WM
        }
        for (int bpc : {4, 8}) {
            const int grid = cus * bpc;
#define WB(L)                                                                                     \
            snprintf(name, sizeof name, "write 64K + %2d KiB read burst / %3d tiles %2d b/CU", L, L * 2, bpc); \
            time_it(name, bytes, [&] { write_burst_kernel<L><<<grid, 256>>>(buf, n_tiles, table, n_entries / 16); });
            WB(1) WB(4) WB(16)
#undef WB
        }
        for (int bpc : {1, 2}) {
            const int grid = cus * bpc;
            if (bpc == 1) {
                snprintf(name, sizeof name, "write phased, 128 KiB LDS burst, 1024 thr, %d b/CU", bpc);
                time_it(name, bytes, [&] { write_phased_kernel<128><<<grid, 1024>>>(buf, n_tiles, table, n_entries / 16); });
            }
            snprintf(name, sizeof name, "write phased,  64 KiB LDS burst, 1024 thr, %d b/CU", bpc);
            time_it(name, bytes, [&] { write_phased_kernel<64><<<grid, 1024>>>(buf, n_tiles, table, n_entries / 16); });
            snprintf(name, sizeof name, "write phased,  16 KiB LDS burst, 1024 thr, %d b/CU", bpc);
            time_it(name, bytes, [&] { write_phased_kernel<16><<<grid, 1024>>>(buf, n_tiles, table, n_entries / 16); });
        }
        // a table small enough to stay in every XCD's L2
        for (int bpc : {4, 8}) {
            const int grid = cus * bpc;
            snprintf(name, sizeof name, "write 64K + 64B/8KiB scalar, 1 MiB table %2d b/CU", bpc);
            time_it(name, bytes, [&] { write_mixed_kernel<8, false><<<grid, 256>>>(buf, n_tiles, table, 1u << 14); });
            snprintf(name, sizeof name, "write 64K + 64B/8KiB scalar, 16 MiB table %2d b/CU", bpc);
            time_it(name, bytes, [&] { write_mixed_kernel<8, false><<<grid, 256>>>(buf, n_tiles, table, 1u << 18); });
        }
        CHECK(hipFree(table));
    }
    for (int bpc : {2, 4, 8}) {
        const int grid = cus * bpc;
        const uint32_t kib = (uint32_t)(bytes >> 10);
#define WT(TK, NTS, LABEL)                                                                        \
        snprintf(name, sizeof name, "write tile %2d KiB %-5s %2d blocks/CU", TK, LABEL, bpc);         \
        time_it(name, bytes, [&] { write_tile_kernel<TK, NTS><<<grid, 256>>>(buf, kib / TK); });
        WT(1, false, "plain") WT(4, false, "plain") WT(16, false, "plain") WT(64, false, "plain")
        WT(1, true, "nt") WT(4, true, "nt") WT(16, true, "nt")
#undef WT
#define RT(TK, UN)                                                                                \
        snprintf(name, sizeof name, "read  tile %2d KiB unroll %d %2d blocks/CU", TK, UN, bpc);       \
        time_it(name, bytes, [&] { read_tile_kernel<TK, UN><<<grid, 256>>>(buf, kib / TK, sink); });
        RT(1, 1) RT(4, 4) RT(8, 8) RT(16, 8) RT(64, 8)
#undef RT
    }
    CHECK(hipFree(buf));
    CHECK(hipFree(buf2));
    CHECK(hipFree(sink));
    return 0;
}
