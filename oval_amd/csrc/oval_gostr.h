// oval_gostr.h — the two Go standard-library string behaviours the host side
// has to reproduce: utf8.RuneCountInString (fmt pads "%-16s" by RUNE count,
// internal/pattern/pattern.go:95,98) and the width of a padded field.
#pragma once
#include <cstddef>
#include <cstdint>

namespace oval {

// utf8.RuneCountInString: an invalid or truncated sequence counts one rune per byte.
inline size_t go_rune_count(const unsigned char* s, size_t n)
{
    size_t cnt = 0, i = 0;
    while (i < n) {
        const unsigned c = s[i];
        size_t sz = 1;
        unsigned lo = 0x80, hi = 0xBF;
        if (c >= 0xC2 && c <= 0xDF) sz = 2;
        else if (c >= 0xE0 && c <= 0xEF) { sz = 3; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
        else if (c >= 0xF0 && c <= 0xF4) { sz = 4; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
        if (sz > 1) {
            bool ok = i + sz <= n && s[i + 1] >= lo && s[i + 1] <= hi;
            for (size_t k = 2; ok && k < sz; ++k) ok = s[i + k] >= 0x80 && s[i + k] <= 0xBF;
            if (!ok) sz = 1;
        }
        i += sz;
        ++cnt;
    }
    return cnt;
}

// byte length of fmt.Sprintf("%-<width>s", s)
inline size_t go_padded_len(const unsigned char* s, size_t n, size_t width)
{
    const size_t runes = go_rune_count(s, n);
    return n + (runes < width ? width - runes : 0);
}

}  // namespace oval
