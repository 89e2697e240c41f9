// pcie_probe.hip — what the GPU's own loads and stores reach on pinned host memory over PCIe, in the
// shapes the small-batch kernel (host_batch_kernel: one workgroup per job, the bodies read from and
// written to pinned memory by the kernel itself) could take: W workgroups of 256 threads, each moving
// a contiguous `per` KiB with U 16-byte accesses per thread in flight; reads, writes, or half of the
// workgroups each.  Reported next to hipMemcpyAsync of the same bytes.  Not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/pcie_probe tools/pcie_probe.hip && build/pcie_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

// mode 0: every workgroup reads; 1: every workgroup writes; 2: even workgroups read, odd ones write
template <int U>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint32_t per16,
                                             int mode, uint32_t* sink)
{
    const bool writes = mode == 1 || (mode == 2 && (blockIdx.x & 1u));
    const size_t base = (size_t)blockIdx.x * per16;
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < per16; i += 256u * U) {
        if (writes) {
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (i + 256u * k < per16)
                    __builtin_nontemporal_store((u32x4){i, (uint32_t)k, 2u, 3u}, dst + base + i + 256u * k);
        } else {
            u32x4 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k)
                v[k] = i + 256u * k < per16 ? __builtin_nontemporal_load(src + base + i + 256u * k) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < U; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int U>
static double run(hipStream_t s, const u32x4* src, u32x4* dst, uint32_t wgs, uint32_t per_bytes, int mode, uint32_t* sink)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        CHECK(hipEventRecord(a, s));
        probe<U><<<wgs, 256, 0, s>>>(src, dst, per_bytes / 16u, mode, sink);
        CHECK(hipEventRecord(b, s));
        CHECK(hipEventSynchronize(b));
        float t;
        CHECK(hipEventElapsedTime(&t, a, b));
        if (it >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return ms[ms.size() / 2];
}

int main()
{
    const size_t cap = 64u << 20;
    void *h_src, *h_dst, *d_buf;
    uint32_t* sink;
    CHECK(hipHostMalloc(&h_src, cap, hipHostMallocCoherent | hipHostMallocMapped));
    CHECK(hipHostMalloc(&h_dst, cap, hipHostMallocCoherent | hipHostMallocMapped));
    CHECK(hipMalloc(&d_buf, cap));
    CHECK(hipMalloc((void**)&sink, 64));
    memset(h_src, 0x5A, cap);
    memset(h_dst, 0, cap);
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    printf("# median of 10 launches; time = HIP events around the launch; GB/s = bytes moved / time\n");
    for (uint32_t per_kib : {8u, 16u, 64u}) {
        for (uint32_t wgs : {32u, 64u, 128u, 256u, 512u}) {
            if ((size_t)wgs * per_kib * 1024 > cap) continue;
            const double bytes = (double)wgs * per_kib * 1024;
            for (int mode = 0; mode < 3; ++mode) {
                const double t1 = run<1>(s, (const u32x4*)h_src, (u32x4*)h_dst, wgs, per_kib * 1024, mode, sink);
                const double t4 = run<4>(s, (const u32x4*)h_src, (u32x4*)h_dst, wgs, per_kib * 1024, mode, sink);
                const double t8 = run<8>(s, (const u32x4*)h_src, (u32x4*)h_dst, wgs, per_kib * 1024, mode, sink);
                printf("%4u workgroups x %3u KiB = %7.0f KB  %-10s  U=1 %7.1f us %5.1f GB/s | U=4 %7.1f us %5.1f GB/s | U=8 %7.1f us %5.1f GB/s\n",
                       wgs, per_kib, bytes / 1e3, mode == 0 ? "read" : mode == 1 ? "write" : "read+write", t1 * 1e3,
                       bytes / t1 / 1e6, t4 * 1e3, bytes / t4 / 1e6, t8 * 1e3, bytes / t8 / 1e6);
            }
        }
    }
    // the copy engines on the same bytes
    for (size_t bytes : {(size_t)256 << 10, (size_t)1 << 20, (size_t)4 << 20}) {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        for (int dir = 0; dir < 2; ++dir) {
            std::vector<float> ms;
            for (int it = 0; it < 12; ++it) {
                CHECK(hipEventRecord(a, s));
                if (dir == 0) CHECK(hipMemcpyAsync(d_buf, h_src, bytes, hipMemcpyHostToDevice, s));
                else CHECK(hipMemcpyAsync(h_dst, d_buf, bytes, hipMemcpyDeviceToHost, s));
                CHECK(hipEventRecord(b, s));
                CHECK(hipEventSynchronize(b));
                float t;
                CHECK(hipEventElapsedTime(&t, a, b));
                if (it >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("hipMemcpyAsync %s %5zu KiB: %7.1f us %5.1f GB/s\n", dir == 0 ? "H2D" : "D2H", bytes >> 10, ms[5] * 1e3,
                   (double)bytes / ms[5] / 1e6);
        }
    }
    return 0;
}
