// stream_bench.cpp — rates of the two large-object host paths against the PCIe
// copies that bound them:
//   * Valid over a reader (internal/pattern/pattern.go:118-134; the body of
//     s3client.GetObject, internal/s3client/s3client.go:238-251) for one big
//     object, streamed through the pinned double buffer, vs a plain pinned
//     host-to-device copy of the same bytes;
//   * GenerateParts (Generate for a multipart upload, s3client.go:164-236) vs a
//     plain device-to-host copy.
// The reader here is memory: one that copies with a single thread (what an
// io.Reader over a []byte does) and one that spreads each copy over several
// threads, so that the reader is not what the measurement shows.
//
//   hipcc -O2 -std=c++17 -o build/stream_bench tools/stream_bench.cpp -Ioval_amd/csrc -Iinclude -Loval_amd -loval_hip -Wl,-rpath,$PWD/oval_amd -lpthread
//
//   stream_bench [MiB of the object, default 256] [repetitions, default 5] [reader threads, default 8]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "oval_pattern.hpp"

using namespace oval;

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// a reader over memory whose Read copies with `threads` threads
class MemReader : public pattern::Reader {
public:
    MemReader(const uint8_t* p, size_t n, int threads) : p_(p), n_(n), threads_(threads) {}
    size_t Read(uint8_t* dst, size_t want) override
    {
        const size_t left = n_ - pos_;
        if (want > left) want = left;
        if (want == 0) return 0;
        if (threads_ <= 1 || want < (4u << 20)) {
            std::memcpy(dst, p_ + pos_, want);
        } else {
            std::vector<std::thread> ts;
            const size_t per = (want / threads_ + 4095) & ~(size_t)4095;
            for (int t = 0; t < threads_; ++t) {
                const size_t a = (size_t)t * per;
                if (a >= want) break;
                const size_t b = a + per < want ? a + per : want;
                ts.emplace_back([=] { std::memcpy(dst + a, p_ + pos_ + a, b - a); });
            }
            for (auto& t : ts) t.join();
        }
        pos_ += want;
        return want;
    }
private:
    const uint8_t* p_;
    size_t n_, pos_ = 0;
    int threads_;
};

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 256;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const int rthreads = argc > 3 ? atoi(argv[3]) : 8;
    const size_t size = mib << 20;
    if (pattern::Init(0)) { std::fprintf(stderr, "init: %s\n", pattern::LastEngineError().c_str()); return 1; }
    pattern::SetClock(1709558129112997ll);
    object::Object obj = object::NewObject(0x20000e2);
    obj.Size = (int64_t)size;
    obj.WriteCount = 2;
    const std::string bucket = "test-bucket";

    // the PCIe copies themselves
    void *pin = nullptr, *dev = nullptr;
    if (hipHostMalloc(&pin, size, hipHostMallocDefault) != hipSuccess || hipMalloc(&dev, size) != hipSuccess) return 2;
    std::memset(pin, 1, size);
    double h2d = 0, d2h = 0;
    for (int r = 0; r < reps + 1; ++r) {
        double t = now_s();
        (void)hipMemcpy(dev, pin, size, hipMemcpyHostToDevice);
        const double a = now_s() - t;
        t = now_s();
        (void)hipMemcpy(pin, dev, size, hipMemcpyDeviceToHost);
        const double b = now_s() - t;
        if (r) { h2d += a; d2h += b; }
    }
    h2d = size * (double)reps / h2d / 1e9;
    d2h = size * (double)reps / d2h / 1e9;
    std::printf("object of %zu MiB, %d repetitions after one warm-up\n", mib, reps);
    std::printf("pinned hipMemcpy             : H2D %6.1f GB/s   D2H %6.1f GB/s\n", h2d, d2h);

    // GenerateParts: part views of 8 MiB, touched (first and last byte) as they arrive
    std::vector<uint8_t> body(size);
    double gen = 0, first_part_ms = 0;
    for (int r = 0; r < reps + 1; ++r) {
        const double t = now_s();
        auto pr = pattern::GenerateParts((int64_t)size, 0x561b, bucket, &obj, 8 << 20);
        if (!pr.second.nil()) { std::fprintf(stderr, "GenerateParts: %s\n", pr.second.msg.c_str()); return 3; }
        pattern::GeneratedParts* parts = pr.first;
        unsigned sink = 0;
        double t_first = 0;
        for (int64_t i = 0; i < parts->Count(); ++i) {
            const uint8_t* p = nullptr;
            size_t n = 0;
            const Error e = parts->Part(i, &p, &n);
            if (!e.nil()) { std::fprintf(stderr, "Part: %s\n", e.msg.c_str()); return 3; }
            if (i == 0) t_first = now_s() - t;
            sink += p[0] + p[n - 1];
            if (r == reps) std::memcpy(body.data() + (size_t)i * (8u << 20), p, n);   // keep the last one for Valid
            if (r != reps) parts->Release(i);
        }
        const double dt = now_s() - t;
        delete parts;
        if (r && r != reps) { gen += dt; first_part_ms += t_first * 1e3; }
        (void)sink;
    }
    const int timed = reps - 1 > 0 ? reps - 1 : 1;
    std::printf("GenerateParts (8 MiB parts)  : %6.1f GB/s = %4.0f %% of D2H; first part after %.2f ms\n",
                size * (double)timed / gen / 1e9, 100.0 * size * timed / gen / 1e9 / d2h, first_part_ms / timed);

    for (int threads : {1, rthreads}) {
        double dt = 0;
        for (int r = 0; r < reps + 1; ++r) {
            MemReader rd(body.data(), size, threads);
            const double t = now_s();
            const Error e = pattern::Valid(0x561b, bucket, &obj, rd);
            const double d = now_s() - t;
            if (!e.nil()) { std::fprintf(stderr, "Valid: %s\n", e.msg.c_str()); return 4; }
            if (r) dt += d;
        }
        std::printf("Valid(reader), %d-thread reader: %6.1f GB/s = %4.0f %% of H2D\n", threads, size * (double)reps / dt / 1e9,
                    100.0 * size * reps / dt / 1e9 / h2d);
    }
    // Generate into the caller's own (pageable) buffer, and Valid over a reader that lends its bytes
    // (bytes.NewReader over a []byte): what the Go binding's Generate / Valid do for a body in memory
    {
        std::vector<uint8_t> out(size);
        double dg = 0, dv = 0;
        for (int r = 0; r < reps + 1; ++r) {
            double t = now_s();
            const Error e = pattern::GenerateInto((int64_t)size, 0x561b, bucket, &obj, out.data());
            const double a = now_s() - t;
            if (!e.nil()) { std::fprintf(stderr, "GenerateInto: %s\n", e.msg.c_str()); return 5; }
            pattern::BytesReader rd(out.data(), size);
            t = now_s();
            const Error v = pattern::Valid(0x561b, bucket, &obj, rd);
            const double b = now_s() - t;
            if (!v.nil()) { std::fprintf(stderr, "Valid(bytes): %s\n", v.msg.c_str()); return 5; }
            if (r) { dg += a; dv += b; }
        }
        if (std::memcmp(out.data(), body.data(), size) != 0) { std::fprintf(stderr, "Generate and GenerateParts differ\n"); return 5; }
        std::printf("Generate into pageable memory: %6.1f GB/s = %4.0f %% of D2H\n", size * (double)reps / dg / 1e9,
                    100.0 * size * reps / dg / 1e9 / d2h);
        std::printf("Valid(bytes reader), pageable: %6.1f GB/s = %4.0f %% of H2D\n", size * (double)reps / dv / 1e9,
                    100.0 * size * reps / dv / 1e9 / h2d);
    }
    // the reader alone, into pinned memory
    for (int threads : {1, rthreads}) {
        double dt = 0;
        for (int r = 0; r < reps + 1; ++r) {
            MemReader rd(body.data(), size, threads);
            const double t = now_s();
            for (size_t o = 0; o < size; o += 32u << 20) rd.Read((uint8_t*)pin + o, 32u << 20);
            if (r) dt += now_s() - t;
        }
        std::printf("reader alone, %d thread(s)     : %6.1f GB/s (memcpy into pinned memory, 32 MiB at a time)\n", threads,
                    size * (double)reps / dt / 1e9);
    }
    (void)hipFree(dev);
    (void)hipHostFree(pin);
    pattern::Shutdown();
    return 0;
}
