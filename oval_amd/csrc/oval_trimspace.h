// oval_trimspace.h — strings.TrimSpace on a fixed-width name field, usable from
// device and host code.
//
// validDataUnit compares strings.TrimSpace(field) with the expected name
// (internal/pattern/pattern.go:140,147), so a field is accepted iff it is
// <white space>* expected <white space>* in Go's sense of white space
// (unicode.IsSpace over UTF-8 decoded runes; an invalid byte is U+FFFD, which
// is not white space).  Every white-space rune has exactly one valid encoding,
// so trimming reduces to matching these byte sequences at either end:
//   09..0D, 20                      ASCII
//   C2 85, C2 A0                    U+0085, U+00A0
//   E1 9A 80                        U+1680
//   E2 80 80..8A, E2 80 A8, E2 80 A9, E2 80 AF, E2 81 9F     U+2000..200A, 2028, 2029, 202F, 205F
//   E3 80 80                        U+3000
// The backward scan only accepts a sequence that lies wholly inside the part
// left by the forward scan, which is what utf8.DecodeLastRuneInString's bounded
// look-back yields.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define OVAL_HD __host__ __device__ inline
#else
#define OVAL_HD inline
#endif

namespace oval {

OVAL_HD bool ascii_space(uint32_t c) { return c == 0x20u || (c >= 0x09u && c <= 0x0Du); }

OVAL_HD bool ws2(uint32_t a, uint32_t b) { return a == 0xC2u && (b == 0x85u || b == 0xA0u); }

OVAL_HD bool ws3(uint32_t a, uint32_t b, uint32_t c)
{
    if (a == 0xE2u) {
        if (b == 0x80u) return (c >= 0x80u && c <= 0x8Au) || c == 0xA8u || c == 0xA9u || c == 0xAFu;
        return b == 0x81u && c == 0x9Fu;
    }
    if (a == 0xE1u) return b == 0x9Au && c == 0x80u;
    if (a == 0xE3u) return b == 0x80u && c == 0x80u;
    return false;
}

// `get(i)` returns byte i of the field, 0 <= i < n.
template <class Get>
OVAL_HD void trim_space(const Get& get, int n, int& start_out, int& stop_out)
{
    int start = 0;
    while (start < n) {
        uint32_t c = get(start);
        if (c < 0x80u) {
            if (!ascii_space(c)) break;
            start += 1;
        } else if (start + 1 < n && ws2(c, get(start + 1))) {
            start += 2;
        } else if (start + 2 < n && ws3(c, get(start + 1), get(start + 2))) {
            start += 3;
        } else {
            break;
        }
    }
    int stop = n;
    while (stop > start) {
        uint32_t c = get(stop - 1);
        if (c < 0x80u) {
            if (!ascii_space(c)) break;
            stop -= 1;
        } else if (stop - 2 >= start && ws2(get(stop - 2), c)) {
            stop -= 2;
        } else if (stop - 3 >= start && ws3(get(stop - 3), get(stop - 2), c)) {
            stop -= 3;
        } else {
            break;
        }
    }
    start_out = start;
    stop_out = stop;
}

// true iff TrimSpace(field[0:n]) == expected[0:exp_len]
template <class Get, class GetExp>
OVAL_HD bool trimmed_equals(const Get& get, int n, const GetExp& exp, int exp_len)
{
    int a, b;
    trim_space(get, n, a, b);
    if (b - a != exp_len) return false;
    for (int k = 0; k < exp_len; ++k)
        if (get(a + k) != exp(k)) return false;
    return true;
}

}  // namespace oval
