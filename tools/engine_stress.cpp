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
// body validated through a reader that delivers it in pieces; the batch forms;
// several calls in two steps under way at once, begun and polled by one thread)
// while another thread switches micro-batching off and on.  Exit code 0 = every
// call gave what it had to.  (tools/sanitize_host.sh builds the library and this
// file with the address and undefined-behaviour sanitizers on the host side; run
// with ASAN_OPTIONS=detect_leaks=0.  The one complaint left is the sanitizer's own
// check inside the ROCm runtime's exit handler, after main has returned.  Under
// the thread sanitizer's address-space layout the HIP runtime finds no device, so
// races are looked for by this run and by review.)
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
                if (kind >= 10 && kind < 20) {
                    // four put-cycles in two steps, all under way at once on this one thread
                    constexpr int kN = 4;
                    object::Object os[kN];
                    std::vector<uint8_t> bodies[kN];
                    pattern::Pending* h[kN];
                    bool hit[kN];
                    for (int q = 0; q < kN; ++q) {
                        os[q] = objs[(i + (uint64_t)q) % objs.size()];
                        os[q].WriteCount += 7 + q;
                        os[q].Size = pattern::DecideSize(4096, 16384).first;
                        bodies[q].assign((size_t)os[q].Size, 0xEE);
                        h[q] = pattern::GenerateBegin(os[q].Size, wid, "test-bucket", &os[q], bodies[q].data());
                    }
                    for (int q = 0; q < kN; ++q) {
                        Error e = Error::Nil();
                        while (!pattern::Poll(h[q], &e)) std::this_thread::yield();
                        if (!e.nil()) wrong++;
                        auto twin = pattern::Generate(os[q].Size, wid, "test-bucket", &os[q]);   // the blocking call's bytes
                        if (!twin.second.nil() || twin.first != bodies[q]) wrong++;
                        hit[q] = (mix(rs) & 1) != 0;
                        if (hit[q]) bodies[q][(size_t)(mix(rs) % (uint64_t)(os[q].Size / 256)) * 256 + 30] ^= 0x10;
                        h[q] = pattern::ValidBegin(wid, "test-bucket", &os[q], bodies[q].data(), bodies[q].size());
                    }
                    for (int q = 0; q < kN; ++q) {
                        Error e = Error::Nil();
                        while (!pattern::Poll(h[q], &e)) std::this_thread::yield();
                        if (e.nil() == hit[q]) wrong++;
                        if (hit[q] && e.msg.find("WriteCount is wrong") == std::string::npos) wrong++;
                    }
                    calls += 3 * kN;
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
    // the run, with a watch on progress: calls that stop completing are reported with the engines' state, not waited for
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
    uint64_t seen = 0;
    int quiet = 0;
    while (std::chrono::steady_clock::now() < t_end) {
        std::this_thread::sleep_for(std::chrono::seconds(1));
        const uint64_t now = calls.load();
        quiet = now == seen ? quiet + 1 : 0;
        seen = now;
        if (quiet >= 10) {
            std::fprintf(stderr, "no call has completed for 10 s (%llu so far)\n", (unsigned long long)now);
            pattern::DebugDump(stderr);
            std::_Exit(3);
        }
    }
    stop = true;
    std::thread watchdog([&] {
        for (int k = 0; k < 60; ++k) {
            std::this_thread::sleep_for(std::chrono::seconds(1));
            if (k == 30) {
                std::fprintf(stderr, "threads still not joined 30 s after the stop\n");
                pattern::DebugDump(stderr);
                std::_Exit(4);
            }
        }
    });
    watchdog.detach();
    for (auto& th : ths) th.join();
    toggler.join();
    pattern::Shutdown();
    std::printf("%llu calls from %d threads, %llu wrong outcomes\n", (unsigned long long)calls.load(), n_threads,
                (unsigned long long)wrong.load());
    std::fflush(stdout);
    return wrong.load() ? 2 : 0;
}
