// compat_bench.cpp — the reference's worker loop without S3: N threads, each
// doing what Worker.Put does with the pattern package (DecideSize, Generate,
// validate-after-put; internal/runner/worker.go:77-109) on its own objects,
// through the C++ host mirror, with and without micro-batching.
//
//   hipcc -O2 -std=c++17 -o build/compat_bench tools/compat_bench.cpp -Ioval_amd/csrc -Loval_amd -loval_hip -Wl,-rpath,$PWD/oval_amd -lpthread
//
//   compat_bench [seconds] [threads,threads,...] [lanes,lanes,...] [blocking 0|1|2 (2 = both)] [skip the unbatched run 0|1]
//                [wake fan-out] [wake roots] [caller spin ns] [completion by flag 1|0] [direct below N callers]
// Environment: COMPAT_SIZES=min,max (object sizes, default 4096,16384); COMPAT_SPIN=ns,callers;
// COMPAT_ASYNC=W: every thread carries W workers and steps them round-robin through GenerateBegin / ValidBegin / Poll
// instead of blocking in Generate / Valid — what a goroutine per worker does on the Go scheduler when the binding yields
// between polls (threads x W logical workers on `threads` OS threads, none of which ever sleeps in the library).
#include <sys/resource.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "oval_pattern.hpp"

using namespace oval;

static int64_t g_min_size = 4096, g_max_size = 16384;
static int g_async_workers = 0;

// One worker of the two-step mode: where it is in its put-cycle.
struct Fiber {
    enum { IDLE, GENERATING, VALIDATING } state = IDLE;
    std::vector<object::Object> objs;
    std::vector<uint8_t> body;
    object::Object* o = nullptr;
    pattern::Pending* pending = nullptr;
    int64_t wid = 0;
    uint64_t i = 0;
};

// One step of a worker; true when a put-cycle has just been completed.
static bool step(Fiber& f, std::atomic<uint64_t>& errors, uint64_t& bytes)
{
    Error e = Error::Nil();
    switch (f.state) {
    case Fiber::IDLE: {
        f.o = &f.objs[f.i++ % f.objs.size()];
        f.o->Size = pattern::DecideSize(g_min_size, g_max_size).first;
        f.o->WriteCount++;
        f.body.resize((size_t)f.o->Size);
        f.pending = pattern::GenerateBegin(f.o->Size, f.wid, "test-bucket", f.o, f.body.data());
        f.state = Fiber::GENERATING;
        return false;
    }
    case Fiber::GENERATING:
        if (!pattern::Poll(f.pending, &e)) return false;
        if (!e.nil()) { errors++; f.state = Fiber::IDLE; return false; }
        f.pending = pattern::ValidBegin(f.wid, "test-bucket", f.o, f.body.data(), f.body.size());
        f.state = Fiber::VALIDATING;
        return false;
    default:
        if (!pattern::Poll(f.pending, &e)) return false;
        if (!e.nil()) errors++;
        bytes += 2 * (uint64_t)f.o->Size;
        f.state = Fiber::IDLE;
        return true;
    }
}

static void run(int n_threads, double seconds, bool batching, int lanes, int blocking)
{
    pattern::SetBatchLanes(lanes, blocking);
    pattern::SetMicroBatching(batching, 0);
    uint64_t b0 = 0, j0 = 0;
    pattern::MicroBatchingStats(&b0, &j0);
    uint64_t prof[8];
    pattern::LaneProfile(prof, true);
    rusage ru0;
    getrusage(RUSAGE_SELF, &ru0);
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> ops{0}, bytes{0}, errors{0};
    std::vector<std::thread> ths;
    for (int t = 0; t < n_threads; ++t) {
        if (g_async_workers > 0 && batching) {
            ths.emplace_back([&, t] {
                std::vector<Fiber> fibers((size_t)g_async_workers);
                for (int w = 0; w < g_async_workers; ++w) {
                    const int id = t * g_async_workers + w;
                    for (int j = 0; j < 64; ++j) fibers[(size_t)w].objs.push_back(object::NewObject(((int64_t)id << 24) + j));
                    fibers[(size_t)w].wid = 0x561B + id;
                }
                uint64_t my_ops = 0, my_bytes = 0;
                while (!stop.load(std::memory_order_relaxed))
                    for (Fiber& f : fibers) my_ops += step(f, errors, my_bytes) ? 1 : 0;
                for (Fiber& f : fibers)                          // every call under way is collected
                    while (f.state != Fiber::IDLE) (void)step(f, errors, my_bytes);
                ops += my_ops;
                bytes += my_bytes;
            });
            continue;
        }
        ths.emplace_back([&, t] {
            std::vector<object::Object> objs;
            for (int j = 0; j < 256; ++j) objs.push_back(object::NewObject(((int64_t)t << 24) + j));
            uint64_t i = 0, my_ops = 0, my_bytes = 0;
            while (!stop.load(std::memory_order_relaxed)) {
                object::Object& o = objs[i++ % objs.size()];
                auto sz = pattern::DecideSize(g_min_size, g_max_size);
                o.Size = sz.first;
                o.WriteCount++;
                auto body = pattern::Generate(o.Size, 0x561B + t, "test-bucket", &o);
                if (!body.second.nil()) { errors++; continue; }
                pattern::BytesReader rd(body.first.data(), body.first.size());
                Error e = pattern::Valid(0x561B + t, "test-bucket", &o, rd);
                if (!e.nil()) errors++;
                my_ops++;
                my_bytes += 2 * (uint64_t)o.Size;
            }
            ops += my_ops;
            bytes += my_bytes;
        });
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto& th : ths) th.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t b = 0, j = 0;
    pattern::MicroBatchingStats(&b, &j);
    if (g_async_workers > 0 && batching) std::printf("two-step calls, %d workers per thread: ", g_async_workers);
    std::printf("threads %3d  batching %-3s lanes %d %-8s: %9.0f put-cycles/s  %8.3f GB/s generate+validate  errors %llu  jobs/launch %.1f\n",
                n_threads, batching ? "on" : "off", lanes, blocking ? "blocking" : "polling", ops / dt, bytes / dt / 1e9,
                (unsigned long long)errors.load(), b > b0 ? (double)(j - j0) / (double)(b - b0) : 0.0);
    rusage ru1;
    getrusage(RUSAGE_SELF, &ru1);
    auto secs_of = [](const timeval& a, const timeval& b) { return (b.tv_sec - a.tv_sec) + 1e-6 * (b.tv_usec - a.tv_usec); };
    const double user = secs_of(ru0.ru_utime, ru1.ru_utime), sys = secs_of(ru0.ru_stime, ru1.ru_stime);
    std::printf("      host CPU: %.1f cores user + %.1f cores system; per put-cycle %.1f us user + %.1f us system; context switches/put-cycle %.1f voluntary + %.2f involuntary\n",
                user / dt, sys / dt, ops ? user / ops * 1e6 : 0.0, ops ? sys / ops * 1e6 : 0.0,
                ops ? (double)(ru1.ru_nvcsw - ru0.ru_nvcsw) / ops : 0.0, ops ? (double)(ru1.ru_nivcsw - ru0.ru_nivcsw) / ops : 0.0);
    pattern::LaneProfile(prof, true);
    if (batching && prof[5])
        std::printf("      per batch, us: callers copying in %.1f | enqueue %.1f | device %.1f | publish+wake %.1f | lock %.1f ; %.0f KB/batch, lanes busy %.0f %%\n",
                    prof[0] / 1e3 / prof[5], prof[1] / 1e3 / prof[5], prof[2] / 1e3 / prof[5], prof[3] / 1e3 / prof[5],
                    prof[4] / 1e3 / prof[5], prof[6] / 1e3 / prof[5],
                    100.0 * (prof[0] + prof[1] + prof[2] + prof[3] + prof[4]) / 1e9 / dt / lanes);
    std::fflush(stdout);
    pattern::SetMicroBatching(false, -1);
}

static std::vector<int> ints(const char* s)
{
    std::vector<int> v;
    for (const char* p = s; *p;) {
        v.push_back(atoi(p));
        p = strchr(p, ',');
        if (!p) break;
        ++p;
    }
    return v;
}

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    const std::vector<int> threads = ints(argc > 2 ? argv[2] : "1,4,16,64,255");
    const std::vector<int> lanes = ints(argc > 3 ? argv[3] : "4");
    const int blocking = argc > 4 ? atoi(argv[4]) : 0;
    const bool skip_off = argc > 5 && atoi(argv[5]) != 0;
    const int fanout = argc > 6 ? atoi(argv[6]) : -1, roots = argc > 7 ? atoi(argv[7]) : -1, spin = argc > 8 ? atoi(argv[8]) : -1;
    if (pattern::Init(0) != 0) { std::fprintf(stderr, "init failed: %s\n", pattern::LastEngineError().c_str()); return 1; }
    pattern::Seed(1);
    pattern::SetWakeTuning(fanout, roots, spin);
    if (const char* e = std::getenv("COMPAT_SPIN")) {          // "<ns>,<callers that may poll at once>"; -1 = the library's default
        int ns = -1, budget = -1;
        std::sscanf(e, "%d,%d", &ns, &budget);
        pattern::SetCallerSpin(ns, budget);
        std::printf("caller spin %d ns, at most %d callers polling (-1 = default)\n", ns, budget);
    }
    std::printf("engines: %d\n", pattern::EngineCount());
    if (const char* e = std::getenv("COMPAT_SIZES")) {
        long long lo = 4096, hi = 16384;
        std::sscanf(e, "%lld,%lld", &lo, &hi);
        g_min_size = lo; g_max_size = hi;
        std::printf("object sizes %lld - %lld bytes\n", lo, hi);
    }
    if (const char* e = std::getenv("COMPAT_ASYNC")) g_async_workers = atoi(e);
    if (argc > 10) {
        pattern::SetDirectCallers(atoi(argv[10]));
        std::printf("calls launch on their own while at most %d are under way\n", atoi(argv[10]));
    }
    if (argc > 9) {
        pattern::SetFlagCompletion(atoi(argv[9]) != 0);
        std::printf("small batches end by %s\n", atoi(argv[9]) ? "a flag their kernel writes to host memory" : "a stream event");
    }
    if (argc > 6) std::printf("wake tree: fan-out %d, roots %d, caller spin %d ns (negative = default)\n", fanout, roots, spin);
    std::printf("host threads available to this process: %u\n", std::thread::hardware_concurrency());
    for (int n : threads) {
        if (!skip_off) run(n, secs, false, 1, 0);
        for (int l : lanes) {
            if (blocking != 1) run(n, secs, true, l, 0);
            if (blocking != 0) run(n, secs, true, l, 1);
        }
    }
    pattern::Shutdown();
    return 0;
}
