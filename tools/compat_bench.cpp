// compat_bench.cpp — the reference's worker loop without S3: N threads, each
// doing what Worker.Put does with the pattern package (DecideSize, Generate,
// validate-after-put; internal/runner/worker.go:77-109) on its own objects,
// through the C++ host mirror, with and without micro-batching.
//
//   hipcc -O2 -std=c++17 -o build/compat_bench tools/compat_bench.cpp -Ioval_amd/csrc -Loval_amd -loval_hip -Wl,-rpath,$PWD/oval_amd -lpthread
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "oval_pattern.hpp"

using namespace oval;

static void run(int n_threads, double seconds, bool batching, int window_us)
{
    pattern::SetMicroBatching(batching, window_us);
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> ops{0}, bytes{0}, errors{0};
    std::vector<std::thread> ths;
    for (int t = 0; t < n_threads; ++t) {
        ths.emplace_back([&, t] {
            std::vector<object::Object> objs;
            for (int j = 0; j < 256; ++j) objs.push_back(object::NewObject(((int64_t)t << 24) + j));
            uint64_t i = 0;
            while (!stop.load(std::memory_order_relaxed)) {
                object::Object& o = objs[i++ % objs.size()];
                auto sz = pattern::DecideSize(4096, 16384);
                o.Size = sz.first;
                o.WriteCount++;
                auto body = pattern::Generate(o.Size, 0x561B + t, "test-bucket", &o);
                if (!body.second.nil()) { errors++; continue; }
                pattern::BytesReader rd(body.first.data(), body.first.size());
                Error e = pattern::Valid(0x561B + t, "test-bucket", &o, rd);
                if (!e.nil()) errors++;
                ops++;
                bytes += 2 * (uint64_t)o.Size;
            }
        });
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto& th : ths) th.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t b = 0, j = 0;
    pattern::MicroBatchingStats(&b, &j);
    std::printf("threads %3d  batching %-3s window %4d us : %9.0f put-cycles/s  %8.3f GB/s generate+validate  errors %llu  (jobs/launch so far %.1f)\n",
                n_threads, batching ? "on" : "off", window_us, ops / dt, bytes / dt / 1e9,
                (unsigned long long)errors.load(), b ? (double)j / b : 0.0);
    pattern::SetMicroBatching(false, -1);
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    if (pattern::Init(0) != 0) { std::fprintf(stderr, "init failed: %s\n", pattern::LastEngineError().c_str()); return 1; }
    pattern::Seed(1);
    for (int n : {1, 4, 16, 64, 255}) {
        run(n, secs, false, 0);
        run(n, secs, true, 0);
        run(n, secs, true, 20);
        run(n, secs, true, 50);
    }
    pattern::Shutdown();
    return 0;
}
