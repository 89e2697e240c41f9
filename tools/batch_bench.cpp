// batch_bench.cpp — GenerateBatch / ValidBatch (many objects per call, bodies in
// the caller's own buffers, one per object) at several batch sizes.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/batch_bench.cpp -Ioval_amd/csrc -Loval_amd -loval_hip \
//       -Wl,-rpath,'$ORIGIN/../oval_amd' -lpthread -o build/batch_bench
//   build/batch_bench [objects,objects,...] [min size] [max size] [sub-batch MiB] [sub-batches in flight]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oval_pattern.hpp"

using namespace oval;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    std::vector<int> counts;
    for (const char* p = argc > 1 ? argv[1] : "64,1024,16384,131072"; p && *p;) {
        counts.push_back(atoi(p));
        p = strchr(p, ',');
        if (p) ++p;
    }
    const int64_t lo = argc > 2 ? atoll(argv[2]) : 4096, hi = argc > 3 ? atoll(argv[3]) : 16384;
    if (pattern::Init(0) != 0) { std::fprintf(stderr, "init failed: %s\n", pattern::LastEngineError().c_str()); return 1; }
    if (argc > 4) pattern::SetBatchShape((size_t)(atof(argv[4]) * (1 << 20)), argc > 5 ? atoi(argv[5]) : 0);
    if (argc > 4) std::printf("sub-batches of %s MiB, %s in flight\n", argv[4], argc > 5 ? argv[5] : "3");
    pattern::Seed(1);
    pattern::SetClock(1709558129112997ll);
    for (int n : counts) {
        std::vector<object::Object> objs;
        std::vector<std::vector<uint8_t>> bodies((size_t)n);
        std::vector<pattern::GenerateItem> gen((size_t)n);
        std::vector<pattern::ValidItem> val((size_t)n);
        uint64_t bytes = 0;
        objs.reserve((size_t)n);
        for (int i = 0; i < n; ++i) {
            objs.push_back(object::NewObject(((int64_t)(i >> 16) << 24) + (i & 0xFFFF)));
            objs.back().Size = pattern::DecideSize(lo, hi).first;
            objs.back().WriteCount = 1 + i % 7;
            bodies[(size_t)i].resize((size_t)objs.back().Size);
            bytes += (uint64_t)objs.back().Size;
        }
        for (int i = 0; i < n; ++i) {
            gen[(size_t)i].dataSize = objs[(size_t)i].Size; gen[(size_t)i].workerID = 0x561B; gen[(size_t)i].bucketName = "test-bucket";
            gen[(size_t)i].obj = &objs[(size_t)i]; gen[(size_t)i].out = bodies[(size_t)i].data();
            val[(size_t)i].workerID = 0x561B; val[(size_t)i].expectedBucketName = "test-bucket"; val[(size_t)i].obj = &objs[(size_t)i];
            val[(size_t)i].body = bodies[(size_t)i].data(); val[(size_t)i].bodyLen = bodies[(size_t)i].size();
        }
        double tg = 1e9, tv = 1e9;
        size_t bad = 0;
        for (int rep = 0; rep < 5; ++rep) {
            double t = now();
            auto ge = pattern::GenerateBatch(gen);
            tg = std::min(tg, now() - t);
            t = now();
            auto ve = pattern::ValidBatch(val);
            tv = std::min(tv, now() - t);
            for (auto& e : ge) bad += !e.nil();
            for (auto& e : ve) bad += !e.nil();
        }
        // a damaged body must be found
        bodies[(size_t)(n / 2)][256 + 29] ^= 1;
        auto ve = pattern::ValidBatch(val);
        size_t found = 0;
        for (auto& e : ve) found += !e.nil();
        std::printf("%7d objects %8.1f MB: GenerateBatch %8.3f ms %7.2f GB/s | ValidBatch %8.3f ms %7.2f GB/s | errors %zu, damaged found %zu\n",
                    n, bytes / 1e6, tg * 1e3, bytes / tg / 1e9, tv * 1e3, bytes / tv / 1e9, bad, found);
        std::fflush(stdout);
    }
    pattern::Shutdown();
    return 0;
}
