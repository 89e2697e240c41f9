// engine_stress.cpp — every host-side path of the pattern engine from many
// threads at once, on a GPU box:
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/engine_stress.cpp -Ioval_amd/csrc -Loval_amd -loval_hip \
//       -Wl,-rpath,'$ORIGIN/../oval_amd' -lpthread -o build/engine_stress
//   build/engine_stress [seconds] [threads]
//
// Workers run the reference's loop shape (DecideSize, Generate, Valid; every so
// often a damaged body, which Valid must refuse with the write-count message; a
// multipart-sized body through GenerateParts, its parts touched out of order; a
// body validated through a reader that delivers it in pieces; the batch forms)
// while another thread switches micro-batching off and on.  Exit code 0 = every
// call gave what it had to.  (A build with -Xarch_host -fsanitize=thread links,
// but the HIP runtime finds no device under the thread sanitizer's address-space
// layout, so races are looked for by this run and by review, not by TSan.)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../oval_amd/csrc/oval_pattern.hpp"

using namespace oval;

namespace {

struct Dribble : pattern::Reader {
    const uint8_t* p;
    size_t n, pos = 0, step;
    Dribble(const uint8_t* data, size_t len, size_t s) : p(data), n(len), step(s) {}
    size_t Read(uint8_t* out, size_t want) override
    {
        size_t k = n - pos;
        if (k > want) k = want;
        if (k > step) k = step;
        std::memcpy(out, p + pos, k);
        pos += k;
        return k;
    }
};

uint64_t mix(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    const int n_threads = argc > 2 ? atoi(argv[2]) : 24;
    if (pattern::Init(0) != 0) { std::fprintf(stderr, "init failed: %s\n", pattern::LastEngineError().c_str()); return 1; }
    pattern::Seed(3);
    pattern::SetClock(1709558129112997ll);
    pattern::SetStreamChunkBytes(1u << 20);
    pattern::SetBatchLanes(2, 0);
    pattern::SetMicroBatching(true, 0);
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> calls{0}, wrong{0};
    std::vector<std::thread> ths;
    for (int t = 0; t < n_threads; ++t) {
        ths.emplace_back([&, t] {
            uint64_t rs = 1000 + (uint64_t)t;
            std::vector<object::Object> objs;
            for (int j = 0; j < 64; ++j) objs.push_back(object::NewObject(((int64_t)t << 24) + j));
            for (uint64_t i = 0; !stop.load(std::memory_order_relaxed); ++i) {
                object::Object& o = objs[i % objs.size()];
                const int64_t wid = 0x561B + t;
                const uint64_t kind = mix(rs) % 64;
                o.WriteCount++;
                if (kind == 0) {
                    // a multipart-sized body in parts, touched in a scrambled order
                    o.Size = (int64_t)(3u << 20) + 256 * (int64_t)(mix(rs) % 4096);
                    auto gp = pattern::GenerateParts(o.Size, wid, "test-bucket", &o, 1 << 20);
                    if (!gp.second.nil()) { wrong++; continue; }
                    std::vector<uint8_t> whole((size_t)o.Size);
                    const int64_t n = gp.first->Count();
                    for (int64_t k = 0; k < n; ++k) {
                        const int64_t part = (k * 5 + 1) % n;                  // n is 3 or 4: a permutation
                        const uint8_t* d = nullptr;
                        size_t len = 0;
                        if (!gp.first->Part(part, &d, &len).nil()) { wrong++; break; }
                        std::memcpy(whole.data() + part * (1 << 20), d, len);
                        gp.first->Release(part);
                    }
                    delete gp.first;
                    Dribble rd(whole.data(), whole.size(), 300000 + (size_t)(mix(rs) % 500000));
                    if (!pattern::Valid(wid, "test-bucket", &o, rd).nil()) wrong++;
                    calls += 2;
                    continue;
                }
                auto sz = pattern::DecideSize(4096, 16384);
                o.Size = sz.first;
                auto body = pattern::Generate(o.Size, wid, "test-bucket", &o);
                if (!body.second.nil()) { wrong++; continue; }
                const bool damage = kind < 8;
                if (damage) body.first[(size_t)(mix(rs) % (uint64_t)(o.Size / 256)) * 256 + 30] ^= 0x10;   // a write-count bit
                pattern::BytesReader rd(body.first.data(), body.first.size());
                const Error e = pattern::Valid(wid, "test-bucket", &o, rd);
                if (e.nil() == damage) wrong++;
                if (damage && e.msg.find("WriteCount is wrong") == std::string::npos) wrong++;
                if (kind == 9) {
                    // the batch forms next to the single calls
                    std::vector<uint8_t> out((size_t)o.Size);
                    pattern::GenerateItem gi;
                    gi.dataSize = o.Size; gi.workerID = wid; gi.bucketName = "test-bucket"; gi.obj = &o; gi.out = out.data();
                    auto ge = pattern::GenerateBatch({gi});
                    if (!ge[0].nil() || (!damage && out != body.first)) wrong++;
                    pattern::ValidItem vi;
                    vi.workerID = wid; vi.expectedBucketName = "test-bucket"; vi.obj = &o; vi.body = out.data(); vi.bodyLen = out.size();
                    if (!pattern::ValidBatch({vi})[0].nil()) wrong++;
                }
                calls += 2;
            }
        });
    }
    std::thread toggler([&] {
        for (int k = 0; !stop.load(std::memory_order_relaxed); ++k) {
            std::this_thread::sleep_for(std::chrono::milliseconds(120));
            pattern::SetMicroBatching((k & 1) != 0, 0);
        }
    });
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto& th : ths) th.join();
    toggler.join();
    pattern::Shutdown();
    std::printf("%llu calls from %d threads, %llu wrong outcomes\n", (unsigned long long)calls.load(), n_threads,
                (unsigned long long)wrong.load());
    return wrong.load() ? 2 : 0;
}
