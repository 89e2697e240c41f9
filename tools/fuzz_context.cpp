// fuzz_context.cpp — mutation fuzzing of the saved-context JSON reader on the
// host (build with the sanitizers; no GPU involved):
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
//       tools/fuzz_context.cpp oval_amd/csrc/oval_context.cpp -Loval_amd -loval_hip -Wl,-rpath,$PWD/oval_amd \
//       -o build/fuzz_context && build/fuzz_context 200000
// Property: LoadSavedContext never crashes or reads out of bounds, and whatever
// it accepts survives Marshal -> Load -> Marshal unchanged.
//   build/fuzz_context 300 big
// mutates a context of several megabytes instead (its workers are then read side
// by side on several threads) and requires the same verdict and the same context
// as the in-order route gives for the same text (also worth a run built with
// -fsanitize=thread instead of address).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../oval_amd/csrc/oval_context.hpp"

static uint64_t s_state = 1;
static uint64_t rnd()
{
    uint64_t z = (s_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv)
{
    const long iters = argc > 1 ? atol(argv[1]) : 100000;
    if (argc > 2 && !strcmp(argv[2], "big")) {
        auto big = oval::runner::NewExecutionContext("http://h:9000", {"bucket-a", "bucket-b"}, 60000, 6, 256, 1024, 3, 0);
        if (!big.second.nil()) { fprintf(stderr, "%s\n", big.second.msg.c_str()); return 1; }
        oval::Error pe = oval::runner::PopulateExisting(big.first, 7, 0.6, 50);
        if (!pe.nil()) { fprintf(stderr, "%s\n", pe.msg.c_str()); return 1; }
        const std::string seed = oval::runner::MarshalContext(big.first);
        static const char* bits[] = {"{", "}", "[", "]", ",", ":", "\"", "\\", "null", " ", "{\"bucketsWithObject\":[",
                                     ",{\"bucketsWithObject\":[]}", "\"zz\":[{\"bucketsWithObject\":[]}],", "\n", "\xFF"};
        long accepted = 0, rejected = 0;
        for (long it = 0; it < iters; ++it) {
            std::string t = seed;
            const int n_mut = (int)(rnd() % 3);                  // 0: the text as written
            for (int m = 0; m < n_mut; ++m) {
                const size_t pos = (size_t)(rnd() % t.size());
                switch (rnd() % 4) {
                case 0: t[pos] = (char)(rnd() & 0xFF); break;
                case 1: t.erase(pos, 1 + (size_t)(rnd() % 8)); break;
                case 2: t.insert(pos, bits[rnd() % (sizeof bits / sizeof bits[0])]); break;
                default: {
                    // at a structural place: just before some worker's opening
                    const size_t w = t.find("{\"bucketsWithObject\":[", pos);
                    if (w != std::string::npos) t.insert(w, bits[rnd() % (sizeof bits / sizeof bits[0])]);
                    break;
                }
                }
            }
            std::string in_order = t;
            const size_t wk = in_order.find("\"workers\":[");
            if (wk != std::string::npos) in_order.insert(wk + 11, " ");
            auto r1 = oval::runner::LoadSavedContext(t.data(), t.size());
            auto r2 = oval::runner::LoadSavedContext(in_order.data(), in_order.size());
            if (r1.second.nil() != r2.second.nil()) {
                fprintf(stderr, "routes disagree at iteration %ld: side by side '%s', in order '%s'\n", it, r1.second.msg.c_str(),
                        r2.second.msg.c_str());
                return 4;
            }
            if (!r1.second.nil()) { ++rejected; continue; }
            ++accepted;
            if (oval::runner::MarshalContext(r1.first) != oval::runner::MarshalContext(r2.first)) {
                fprintf(stderr, "routes give different contexts at iteration %ld\n", it);
                return 5;
            }
        }
        printf("%ld large inputs: %ld accepted by both routes alike, %ld rejected by both\n", iters, accepted, rejected);
        return 0;
    }
    auto fresh = oval::runner::NewExecutionContext("http://h:9000/\xC3\xA9<&>", {"bucket-a", "b\"\\\n"}, 12, 3, 256, 1024,
                                                   0xFFFF, 2);
    if (!fresh.second.nil()) { fprintf(stderr, "%s\n", fresh.second.msg.c_str()); return 1; }
    oval::Error e = oval::runner::PopulateExisting(fresh.first, 7, 0.6, 50);
    if (!e.nil()) { fprintf(stderr, "%s\n", e.msg.c_str()); return 1; }
    const std::string seed = oval::runner::MarshalContext(fresh.first);
    static const char* bits[] = {"{", "}", "[", "]", ",", ":", "\"", "\\", "null", "true", "-", "1e9", "0.5", " ",
                                 "\\u", "\\ud800", "\\udc00\\ud800", "\"key\"", "99999999999999999999", "\n", "\x01", "\xFF"};
    long accepted = 0, rejected = 0;
    for (long it = 0; it < iters; ++it) {
        std::string t = seed;
        const int n_mut = 1 + (int)(rnd() % 4);
        for (int m = 0; m < n_mut && !t.empty(); ++m) {
            const size_t pos = (size_t)(rnd() % t.size());
            switch (rnd() % 6) {
            case 0: t[pos] = (char)(rnd() & 0xFF); break;
            case 1: t.erase(pos, 1 + (size_t)(rnd() % 8)); break;
            case 2: t.insert(pos, bits[rnd() % (sizeof bits / sizeof bits[0])]); break;
            case 3: t.resize(pos); break;
            case 4: { const size_t q = (size_t)(rnd() % t.size()); t.insert(pos, t.substr(q, (size_t)(rnd() % 40))); break; }
            default: t[pos] = bits[rnd() % 7][0]; break;
            }
        }
        auto r = oval::runner::LoadSavedContext(t.data(), t.size());
        if (!r.second.nil()) { ++rejected; continue; }
        ++accepted;
        const std::string a = oval::runner::MarshalContext(r.first);
        auto r2 = oval::runner::LoadSavedContext(a.data(), a.size());
        if (!r2.second.nil()) { fprintf(stderr, "re-load failed: %s\n%s\n", r2.second.msg.c_str(), a.c_str()); return 2; }
        if (oval::runner::MarshalContext(r2.first) != a) { fprintf(stderr, "round trip differs\n%s\n", a.c_str()); return 3; }
        auto d = oval::runner::BuildExistingDescriptors(r.first, 1, 0, r.first.Workers.size());
        (void)d;
    }
    printf("%ld inputs: %ld accepted (round trip ok), %ld rejected\n", iters, accepted, rejected);
    return 0;
}
