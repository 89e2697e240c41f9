// oval_context.cpp — ObjectMeta, ExecutionContext (JSON in the reference's
// format) and descriptor tables built from a context.  See oval_context.hpp.
#include "oval_context.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <thread>

namespace oval {

// ---------------------------------------------------------------------------
// rand.Intn
// ---------------------------------------------------------------------------
namespace {

std::mutex g_rand_mu;
uint64_t g_rand_state = 0x9E3779B97F4A7C15ull;

uint64_t splitmix64()
{
    uint64_t z = (g_rand_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// uniform in [0, n), n > 0, without modulo bias
int64_t rand_intn(int64_t n)
{
    std::lock_guard<std::mutex> lk(g_rand_mu);
    const uint64_t un = (uint64_t)n;
    const uint64_t limit = UINT64_MAX - (UINT64_MAX % un + 1) % un;
    uint64_t x;
    do x = splitmix64(); while (x > limit);
    return (int64_t)(x % un);
}

// Tasks 0 .. n-1 on up to eight host threads (the caller is one of them); `weight`
// is the number of objects the tasks cover together: a thread per ~50 K of them.
// A saved context of a million objects is 58 MB of JSON: built, written and read
// on one core it costs more than validating every body it names on the device.
template <class Fn>
void parallel_tasks(size_t n, size_t weight, const Fn& fn)
{
    size_t n_threads = weight / 50000;
    const size_t hw = std::thread::hardware_concurrency();
    if (n_threads > 8) n_threads = 8;
    if (hw && n_threads > hw) n_threads = hw;
    if (n_threads > n) n_threads = n;
    if (n_threads <= 1) {
        for (size_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n;) fn(i);
    };
    std::vector<std::thread> ts;
    for (size_t t = 1; t < n_threads; ++t) ts.emplace_back(work);
    work();
    for (std::thread& t : ts) t.join();
}

}  // namespace

namespace object {

void SeedRandom(uint64_t seed)
{
    std::lock_guard<std::mutex> lk(g_rand_mu);
    g_rand_state = seed;
}

std::pair<int64_t, Error> getObjIDFromKey(const std::string& key)
{
    // strconv.ParseInt(key[KeyPrefixLength:], 16, 64): an optional sign, then hex digits
    if (key.size() <= (size_t)KeyPrefixLength)
        return {0, Error::New("strconv.ParseInt: parsing \"" + (key.size() < (size_t)KeyPrefixLength ? std::string() : key.substr(KeyPrefixLength)) + "\": invalid syntax")};
    const std::string s = key.substr(KeyPrefixLength);
    size_t i = 0;
    bool neg = false;
    if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; i = 1; }
    if (i == s.size()) return {0, Error::New("strconv.ParseInt: parsing \"" + s + "\": invalid syntax")};
    uint64_t v = 0;
    for (; i < s.size(); ++i) {
        const char c = s[i];
        int d;
        if (c >= '0' && c <= '9') d = c - '0';
        else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
        else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
        else return {0, Error::New("strconv.ParseInt: parsing \"" + s + "\": invalid syntax")};
        if (v >> 59) return {0, Error::New("strconv.ParseInt: parsing \"" + s + "\": value out of range")};
        v = v * 16 + (uint64_t)d;
    }
    if (v > (uint64_t)INT64_MAX) return {0, Error::New("strconv.ParseInt: parsing \"" + s + "\": value out of range")};
    return {neg ? -(int64_t)v : (int64_t)v, Error::Nil()};
}

ObjectMeta::ObjectMeta(int64_t numObj, int64_t keyIDOffset)
{
    if (numObj < 0) numObj = 0;
    ObjectList.reserve((size_t)numObj);
    for (int64_t objID = 0; objID < numObj; ++objID) ObjectList.push_back(NewObject(keyIDOffset + objID));
    ExistingObjectIDs.reserve((size_t)std::sqrt((double)numObj));
    existing_.assign((size_t)numObj, 0);
    KeyIDOffset = keyIDOffset;
    KeyPrefix = generateKey(keyIDOffset).substr(0, KeyPrefixLength);
}

Object* ObjectMeta::GetRandomObject()
{
    if (ObjectList.empty()) return nullptr;                      // rand.Intn(0) panics in the reference
    return &ObjectList[(size_t)rand_intn((int64_t)ObjectList.size())];
}

Error ObjectMeta::RegisterToExistingList(const std::string& key)
{
    auto [objID, err] = getObjIDFromKey(key);
    if (!err.nil()) return err;
    if (objID < 0 || (uint64_t)objID >= ObjectList.size())
        return Error::New("object ID of key " + key + " is outside this ObjectMeta's object list");
    if (existing_[(size_t)objID]) return Error::Nil();           // The key is already registered.
    ExistingObjectIDs.push_back(objID);
    existing_[(size_t)objID] = 1;
    if (ObjectList.size() < ExistingObjectIDs.size())
        return Error::New("Invalid contents of existing object ID list.");
    return Error::Nil();
}

Object* ObjectMeta::PopExistingRandomObject()
{
    if (ExistingObjectIDs.empty()) return nullptr;
    const size_t idx = (size_t)rand_intn((int64_t)ExistingObjectIDs.size());
    const int64_t objID = ExistingObjectIDs[idx];
    // Delete the idx-th entry from the existing object ID list
    ExistingObjectIDs[idx] = ExistingObjectIDs.back();
    ExistingObjectIDs.pop_back();
    existing_[(size_t)objID] = 0;
    return &ObjectList[(size_t)objID];
}

Object* ObjectMeta::GetExistingRandomObject()
{
    if (ExistingObjectIDs.empty()) return nullptr;
    const size_t idx = (size_t)rand_intn((int64_t)ExistingObjectIDs.size());
    return &ObjectList[(size_t)ExistingObjectIDs[idx]];
}

std::pair<bool, Error> ObjectMeta::Exist(const std::string& key) const
{
    auto [objID, err] = getObjIDFromKey(key);
    if (!err.nil()) return {false, err};
    if (objID < 0 || (uint64_t)objID >= existing_.size()) return {false, Error::Nil()};
    return {existing_[(size_t)objID] != 0, Error::Nil()};
}

std::pair<std::string, std::string> ObjectMeta::GetHeadAndTailKey() const
{
    if (ObjectList.empty()) return {std::string(), std::string()};
    return {ObjectList.front().Key, ObjectList.back().Key};
}

Error ObjectMeta::TidyUp()
{
    existing_.assign(ObjectList.size(), 0);
    for (int64_t objID : ExistingObjectIDs) {
        if (objID < 0 || (uint64_t)objID >= ObjectList.size())
            return Error::New("existingObjectIDs holds " + std::to_string(objID) + ", outside the object list of " +
                              std::to_string(ObjectList.size()) + " objects");
        existing_[(size_t)objID] = 1;
    }
    return Error::Nil();
}

}  // namespace object

namespace runner {

// ---------------------------------------------------------------------------
// JSON input: a recursive-descent reader that fills the structs directly
// ---------------------------------------------------------------------------
namespace {

// Length of the valid UTF-8 sequence at p (n bytes available), 0 if there is
// none (utf8.DecodeRune's RuneError of width 1).
int utf8_len(const char* p, size_t n, uint32_t& cp)
{
    const unsigned char c = (unsigned char)p[0];
    int len = 0;
    if (c < 0x80) { cp = c; return 1; }
    if (c >= 0xC2 && c <= 0xDF) { len = 2; cp = c & 0x1Fu; }
    else if (c >= 0xE0 && c <= 0xEF) { len = 3; cp = c & 0x0Fu; }
    else if (c >= 0xF0 && c <= 0xF4) { len = 4; cp = c & 0x07u; }
    if (len == 0 || (size_t)len > n) return 0;
    for (int k = 1; k < len; ++k) {
        const unsigned char d = (unsigned char)p[k];
        if ((d & 0xC0) != 0x80) return 0;
        cp = (cp << 6) | (d & 0x3Fu);
    }
    if (len == 3 && (cp < 0x800 || (cp >= 0xD800 && cp < 0xE000))) return 0;
    if (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)) return 0;
    return len;
}

struct Reader {
    const char* p;
    const char* end;
    const char* begin;
    std::string err;

    bool fail(const std::string& m)
    {
        if (err.empty()) err = m + " (at offset " + std::to_string((size_t)(p - begin)) + ")";
        return false;
    }
    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool lit(char c)
    {
        ws();
        if (p < end && *p == c) { ++p; return true; }
        return false;
    }
    bool expect(char c)
    {
        if (lit(c)) return true;
        if (p >= end) return fail("unexpected end of JSON input");
        return fail(std::string("invalid character, expected '") + c + "'");
    }
    bool word(const char* w)
    {
        const size_t n = std::strlen(w);
        if ((size_t)(end - p) >= n && std::memcmp(p, w, n) == 0) { p += n; return true; }
        return false;
    }
    bool null()
    {
        ws();
        return word("null");
    }

    static void utf8(std::string& out, uint32_t c)
    {
        if (c < 0x80) out.push_back((char)c);
        else if (c < 0x800) { out.push_back((char)(0xC0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3F))); }
        else if (c < 0x10000) {
            out.push_back((char)(0xE0 | (c >> 12)));
            out.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (c & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (c >> 18)));
            out.push_back((char)(0x80 | ((c >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (c & 0x3F)));
        }
    }
    bool hex4(uint32_t& v)
    {
        if (end - p < 4) return fail("invalid \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return fail("invalid \\u escape");
        }
        return true;
    }
    bool string(std::string& out)
    {
        ws();
        if (p >= end) return fail("unexpected end of JSON input");
        if (*p != '"') return fail("invalid character, expected a string");
        ++p;
        out.clear();
        while (p < end) {
            const unsigned char c = (unsigned char)*p++;
            if (c == '"') return true;
            if (c < 0x20) return fail("invalid control character in string literal");
            if (c >= 0x80) {
                // encoding/json replaces invalid UTF-8 inside a string with U+FFFD
                uint32_t cp;
                const int len = utf8_len(p - 1, (size_t)(end - (p - 1)), cp);
                if (len == 0) { out += "\xEF\xBF\xBD"; continue; }
                out.append(p - 1, (size_t)len);
                p += len - 1;
                continue;
            }
            if (c != '\\') { out.push_back((char)c); continue; }
            if (p >= end) break;
            const char e = *p++;
            switch (e) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                uint32_t u;
                if (!hex4(u)) return false;
                if (u >= 0xD800 && u < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                    const char* save = p;
                    p += 2;
                    uint32_t lo;
                    if (!hex4(lo)) return false;
                    if (lo >= 0xDC00 && lo < 0xE000) u = 0x10000 + ((u - 0xD800) << 10) + (lo - 0xDC00);
                    else { p = save; u = 0xFFFD; }
                } else if (u >= 0xD800 && u < 0xE000) {
                    u = 0xFFFD;                                 // a lone surrogate decodes to U+FFFD
                }
                utf8(out, u);
                break;
            }
            default: return fail("invalid escape in string literal");
            }
        }
        return fail("unexpected end of JSON input");
    }
    // a JSON number destined for a Go int / int64
    bool integer(int64_t& out, const char* field)
    {
        ws();
        const char* s = p;
        if (p < end && *p == '-') ++p;
        if (p >= end || *p < '0' || *p > '9') { p = s; return fail(std::string("cannot unmarshal into field ") + field + " of type int"); }
        // up to 18 digits cannot overflow: no detour through strtoll for the millions of small numbers of a context
        uint64_t acc = 0;
        while (p < end && *p >= '0' && *p <= '9' && p - s < 18) acc = acc * 10u + (uint64_t)(*p++ - '0');
        if (p >= end || ((*p < '0' || *p > '9') && *p != '.' && *p != 'e' && *p != 'E')) {
            out = *s == '-' ? -(int64_t)acc : (int64_t)acc;
            return true;
        }
        while (p < end && *p >= '0' && *p <= '9') ++p;
        if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) {
            p = s;
            return fail(std::string("cannot unmarshal a non-integer number into field ") + field + " of type int");
        }
        errno = 0;
        const std::string t(s, (size_t)(p - s));
        char* e = nullptr;
        const long long v = std::strtoll(t.c_str(), &e, 10);
        if (errno == ERANGE) { p = s; return fail(std::string("number ") + t + " overflows field " + field); }
        out = (int64_t)v;
        return true;
    }
    // skips any JSON value (an unknown field's)
    bool skip(int depth = 0)
    {
        if (depth > 512) return fail("exceeded max depth");
        ws();
        if (p >= end) return fail("unexpected end of JSON input");
        const char c = *p;
        if (c == '"') { std::string t; return string(t); }
        if (c == '{' || c == '[') {
            const char close = c == '{' ? '}' : ']';
            ++p;
            if (lit(close)) return true;
            for (;;) {
                if (c == '{') {
                    std::string k;
                    if (!string(k) || !expect(':')) return false;
                }
                if (!skip(depth + 1)) return false;
                if (lit(',')) continue;
                return expect(close);
            }
        }
        if (word("true") || word("false") || word("null")) return true;
        if (c == '-' || (c >= '0' && c <= '9')) {
            ++p;
            while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) ++p;
            return true;
        }
        return fail("invalid character looking for beginning of value");
    }
};

// encoding/json matches field names exactly or under simple case folding
bool key_is(const std::string& k, const char* name)
{
    const size_t n = std::strlen(name);
    if (k.size() != n) return false;
    for (size_t i = 0; i < n; ++i) {
        char a = k[i], b = name[i];
        if (a >= 'A' && a <= 'Z') a = (char)(a - 'A' + 'a');
        if (b >= 'A' && b <= 'Z') b = (char)(b - 'A' + 'a');
        if (a != b) return false;
    }
    return true;
}

// `{ "k": v, ... }` with one callback per member; the callback consumes v.
template <class F>
bool read_object(Reader& r, F member)
{
    if (!r.expect('{')) return false;
    if (r.lit('}')) return true;
    std::string k;
    for (;;) {
        if (!r.string(k) || !r.expect(':')) return false;
        if (!member(k)) return false;
        if (r.lit(',')) continue;
        return r.expect('}');
    }
}

// `[ v, ... ]` (or null = empty) with one callback per element
template <class F>
bool read_array(Reader& r, F element)
{
    if (r.null()) return true;
    if (!r.expect('[')) return false;
    if (r.lit(']')) return true;
    for (;;) {
        if (!element()) return false;
        if (r.lit(',')) continue;
        return r.expect(']');
    }
}

bool read_string_field(Reader& r, std::string& dst)
{
    if (r.null()) return true;
    return r.string(dst);
}

bool read_int_field(Reader& r, int64_t& dst, const char* name)
{
    if (r.null()) return true;
    return r.integer(dst, name);
}

// `{"key":"...","size":N,"writeCount":N}` exactly as encoding/json writes an Object (struct order, no
// white space, a key without escapes): the shape of all but a handful of a saved context's bytes.  Anything
// else is left to the general reader.
bool read_object_entry_compact(Reader& r, object::Object& o)
{
    auto starts = [&](const char* q, const char* lit, size_t n) { return (size_t)(r.end - q) >= n && std::memcmp(q, lit, n) == 0; };
    auto number = [&](const char*& q, int64_t& v) {
        const char* s = q;
        const bool neg = q < r.end && *q == '-';
        if (neg) ++q;
        const char* d = q;
        uint64_t acc = 0;
        while (q < r.end && *q >= '0' && *q <= '9' && q - d < 18) acc = acc * 10u + (uint64_t)(*q++ - '0');
        if (q == d || (q < r.end && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E'))) { q = s; return false; }
        v = neg ? -(int64_t)acc : (int64_t)acc;
        return true;
    };
    const char* q = r.p;
    if (!starts(q, "{\"key\":\"", 8)) return false;
    q += 8;
    const char* k = q;
    while (q < r.end && *q != '"' && *q != '\\' && (unsigned char)*q >= 0x20 && (unsigned char)*q < 0x80) ++q;
    if (q >= r.end || *q != '"') return false;
    const size_t klen = (size_t)(q - k);
    ++q;
    int64_t size = 0, wc = 0;
    if (!starts(q, ",\"size\":", 8)) return false;
    q += 8;
    if (!number(q, size)) return false;
    if (!starts(q, ",\"writeCount\":", 14)) return false;
    q += 14;
    if (!number(q, wc)) return false;
    if (q >= r.end || *q != '}') return false;
    r.p = q + 1;
    o.Key.assign(k, klen);
    o.Size = size;
    o.WriteCount = wc;
    return true;
}

bool read_object_entry(Reader& r, object::Object& o)
{
    r.ws();
    if (read_object_entry_compact(r, o)) return true;
    return read_object(r, [&](const std::string& k) {
        if (key_is(k, "key")) return read_string_field(r, o.Key);
        if (key_is(k, "size")) return read_int_field(r, o.Size, "Object.size");
        if (key_is(k, "writeCount")) return read_int_field(r, o.WriteCount, "Object.writeCount");
        return r.skip();
    });
}

bool read_object_meta(Reader& r, object::ObjectMeta& om)
{
    return read_object(r, [&](const std::string& k) {
        if (key_is(k, "objectList")) {
            om.ObjectList.clear();
            return read_array(r, [&] {
                if (r.null()) return r.fail("objectList holds null: the reference would dereference a nil *Object");
                om.ObjectList.emplace_back();
                return read_object_entry(r, om.ObjectList.back());
            });
        }
        if (key_is(k, "existingObjectIDs")) {
            om.ExistingObjectIDs.clear();
            return read_array(r, [&] {
                int64_t v = 0;
                if (!read_int_field(r, v, "ObjectMeta.existingObjectIDs")) return false;
                om.ExistingObjectIDs.push_back(v);
                return true;
            });
        }
        if (key_is(k, "keyIDOffset")) return read_int_field(r, om.KeyIDOffset, "ObjectMeta.keyIDOffset");
        if (key_is(k, "KeyPrefix")) return read_string_field(r, om.KeyPrefix);
        return r.skip();
    });
}

bool read_worker(Reader& r, Worker& w)
{
    return read_object(r, [&](const std::string& k) {
        if (key_is(k, "bucketsWithObject")) {
            w.BucketsWithObject.clear();
            return read_array(r, [&] {
                if (r.null()) return r.fail("bucketsWithObject holds null: the reference would dereference a nil *BucketWithObject");
                w.BucketsWithObject.emplace_back();
                BucketWithObject& b = w.BucketsWithObject.back();
                return read_object(r, [&](const std::string& kk) {
                    if (key_is(kk, "bucketName")) return read_string_field(r, b.BucketName);
                    if (key_is(kk, "objectMeta")) {
                        if (r.null()) return r.fail("objectMeta is null: the reference would dereference a nil *ObjectMeta");
                        return read_object_meta(r, b.ObjectMeta);
                    }
                    return r.skip();
                });
            });
        }
        return r.skip();
    });
}

// The workers array of a large saved context, its elements read side by side.
// encoding/json writes every worker as `{"bucketsWithObject":[...]}` with no
// white space, and a quote inside a JSON string is always escaped, so those
// opening bytes can only occur where an object with that first key begins.  Each
// occurrence is taken as the start of one element and the text up to the next
// one is read as exactly one worker followed by a comma.  Anything that does not
// fit (white space, other key order, such an object nested somewhere else, a
// syntax error) makes this return false with nothing consumed, and the caller
// reads the array in order, element by element, as it would have anyway.
bool read_workers_side_by_side(Reader& r, std::vector<Worker>& out)
{
    static const char kOpen[] = "{\"bucketsWithObject\":[";
    const size_t kOpenLen = sizeof kOpen - 1;
    r.ws();
    const char* const p0 = r.p;
    if ((size_t)(r.end - p0) < (4u << 20) || *p0 != '[') return false;
    std::vector<const char*> starts;
    for (const char* q = p0 + 1; q < r.end;) {
        const char* f = static_cast<const char*>(memmem(q, (size_t)(r.end - q), kOpen, kOpenLen));
        if (!f) break;
        if (f[-1] != (starts.empty() ? '[' : ',')) return false;
        starts.push_back(f);
        q = f + kOpenLen;
    }
    if (starts.size() < 2 || starts[0] != p0 + 1) return false;
    const size_t n = starts.size();
    std::vector<Worker> workers(n);
    std::vector<const char*> stop(n, nullptr);
    std::atomic<bool> ok{true};
    parallel_tasks(n, (size_t)(r.end - p0) / 56, [&](size_t i) {
        if (!ok.load(std::memory_order_relaxed)) return;
        // all but the last element end where the comma before the next one stands
        Reader part{starts[i], i + 1 < n ? starts[i + 1] - 1 : r.end, r.begin, std::string()};
        if (!read_worker(part, workers[i]) || (i + 1 < n && part.p != part.end)) ok.store(false, std::memory_order_relaxed);
        stop[i] = part.p;
    });
    if (!ok.load()) return false;
    r.p = stop[n - 1];
    if (!r.lit(']')) { r.p = p0; return false; }
    out = std::move(workers);
    return true;
}

// ---------------------------------------------------------------------------
// JSON output: encoding/json's compact form with its default HTML escaping
// ---------------------------------------------------------------------------
void put_string(std::string& out, const std::string& s)
{
    static const char* hex = "0123456789abcdef";
    out.push_back('"');
    for (size_t i = 0; i < s.size();) {
        const unsigned char c = (unsigned char)s[i];
        if (c < 0x80) {
            if (c == '"' || c == '\\') { out.push_back('\\'); out.push_back((char)c); }
            else if (c == '\n') out += "\\n";
            else if (c == '\r') out += "\\r";
            else if (c == '\t') out += "\\t";
            // encoding/json of Go >= 1.22 (the reference's go.mod says 1.24) writes \b and \f in
            // their short forms; Go <= 1.21 wrote \u0008 and \u000c.  Readers accept both.
            else if (c == '\b') out += "\\b";
            else if (c == '\f') out += "\\f";
            else if (c < 0x20 || c == '<' || c == '>' || c == '&') {
                out += "\\u00";
                out.push_back(hex[c >> 4]);
                out.push_back(hex[c & 0xF]);
            } else out.push_back((char)c);
            ++i;
            continue;
        }
        // a multi-byte sequence: copy it if it is valid UTF-8, else U+FFFD for one byte
        uint32_t cp = 0;
        const int n = utf8_len(s.data() + i, s.size() - i, cp);
        const bool ok = n != 0;
        if (!ok) { out += "\\ufffd"; ++i; continue; }
        if (cp == 0x2028 || cp == 0x2029) { out += cp == 0x2028 ? "\\u2028" : "\\u2029"; }
        else out.append(s, i, (size_t)n);
        i += (size_t)n;
    }
    out.push_back('"');
}

// decimal digits of v at q; returns the end
char* put_digits(char* q, int64_t v)
{
    uint64_t u = v < 0 ? 0u - (uint64_t)v : (uint64_t)v;
    if (v < 0) *q++ = '-';
    char tmp[20];
    int n = 0;
    do { tmp[n++] = (char)('0' + u % 10u); u /= 10u; } while (u);
    while (n) *q++ = tmp[--n];
    return q;
}

void put_int(std::string& out, int64_t v)
{
    char buf[24];
    out.append(buf, (size_t)(put_digits(buf, v) - buf));
}

// true when put_string would copy s unchanged between its quotes
bool plain_text(const std::string& s)
{
    for (const char ch : s) {
        const unsigned char c = (unsigned char)ch;
        if (c < 0x20 || c >= 0x80 || c == '"' || c == '\\' || c == '<' || c == '>' || c == '&') return false;
    }
    return true;
}

// `{"key":"...","size":N,"writeCount":N}`: the bulk of a saved context, assembled in one go
void put_object_entry(std::string& out, const object::Object& o)
{
    if (o.Key.size() > 64 || !plain_text(o.Key)) {
        out += "{\"key\":";
        put_string(out, o.Key);
        out += ",\"size\":"; put_int(out, o.Size);
        out += ",\"writeCount\":"; put_int(out, o.WriteCount);
        out.push_back('}');
        return;
    }
    char buf[8 + 64 + 9 + 21 + 15 + 21 + 1];
    char* q = buf;
    std::memcpy(q, "{\"key\":\"", 8); q += 8;
    std::memcpy(q, o.Key.data(), o.Key.size()); q += o.Key.size();
    std::memcpy(q, "\",\"size\":", 9); q += 9;
    q = put_digits(q, o.Size);
    std::memcpy(q, ",\"writeCount\":", 14); q += 14;
    q = put_digits(q, o.WriteCount);
    *q++ = '}';
    out.append(buf, (size_t)(q - buf));
}

size_t total_objects(const ExecutionContext& ec)
{
    size_t n = 0;
    for (const Worker& w : ec.Workers)
        for (const BucketWithObject& b : w.BucketsWithObject) n += b.ObjectMeta.ObjectList.size();
    return n;
}

void marshal_worker(std::string& o, const Worker& wk)
{
    size_t objects = 0, ids = 0;
    for (const BucketWithObject& bo : wk.BucketsWithObject) {
        objects += bo.ObjectMeta.ObjectList.size();
        ids += bo.ObjectMeta.ExistingObjectIDs.size();
    }
    o.reserve(256 + objects * 56 + ids * 8);
    o += "{\"bucketsWithObject\":[";
    for (size_t b = 0; b < wk.BucketsWithObject.size(); ++b) {
        if (b) o.push_back(',');
        const BucketWithObject& bo = wk.BucketsWithObject[b];
        o += "{\"bucketName\":";
        put_string(o, bo.BucketName);
        o += ",\"objectMeta\":{\"objectList\":[";
        const object::ObjectMeta& om = bo.ObjectMeta;
        for (size_t i = 0; i < om.ObjectList.size(); ++i) {
            if (i) o.push_back(',');
            put_object_entry(o, om.ObjectList[i]);
        }
        o += "],\"existingObjectIDs\":[";
        for (size_t i = 0; i < om.ExistingObjectIDs.size(); ++i) {
            if (i) o.push_back(',');
            put_int(o, om.ExistingObjectIDs[i]);
        }
        o += "],\"keyIDOffset\":"; put_int(o, om.KeyIDOffset);
        o += ",\"KeyPrefix\":";
        put_string(o, om.KeyPrefix);
        o += "}}";
    }
    o += "]}";
}

}  // namespace

namespace {

// The text of a context in pieces, in order: everything up to and including the
// opening bracket of the workers array, one piece per worker (commas between
// them are the writer's business), and the closing "]}".
void marshal_pieces(const ExecutionContext& ec, std::string& head, std::vector<std::string>& workers)
{
    std::string& o = head;
    o += "{\"endpoint\":";
    put_string(o, ec.Endpoint);
    o += ",\"bucketNames\":";
    if (ec.BucketNames.empty()) o += "[]";
    else {
        o.push_back('[');
        for (size_t i = 0; i < ec.BucketNames.size(); ++i) {
            if (i) o.push_back(',');
            put_string(o, ec.BucketNames[i]);
        }
        o.push_back(']');
    }
    o += ",\"numObj\":"; put_int(o, ec.NumObj);
    o += ",\"numWorker\":"; put_int(o, ec.NumWorker);
    o += ",\"minSize\":"; put_int(o, ec.MinSize);
    o += ",\"maxSize\":"; put_int(o, ec.MaxSize);
    o += ",\"startWorkerID\":"; put_int(o, ec.StartWorkerID);
    o += ",\"workers\":[";
    // every worker's text on its own, side by side
    workers.assign(ec.Workers.size(), std::string());
    parallel_tasks(ec.Workers.size(), total_objects(ec), [&](size_t w) { marshal_worker(workers[w], ec.Workers[w]); });
}

}  // namespace

std::string MarshalContext(const ExecutionContext& ec)
{
    std::string o;
    std::vector<std::string> parts;
    marshal_pieces(ec, o, parts);
    size_t bytes = o.size() + 2;
    for (const std::string& part : parts) bytes += part.size() + 1;
    o.reserve(bytes);
    for (size_t w = 0; w < parts.size(); ++w) {
        if (w) o.push_back(',');
        o += parts[w];
        std::string().swap(parts[w]);
    }
    o += "]}";
    return o;
}

Error SaveContext(const ExecutionContext& ec, const std::string& path)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return Error::New("open " + path + ": " + std::strerror(errno));
    std::string head;
    std::vector<std::string> parts;
    marshal_pieces(ec, head, parts);
    // the pieces go to the file as they are: no second copy of the whole text
    auto put = [&](const char* p, size_t n) {
        while (n) {
            const size_t w = std::fwrite(p, 1, n, f);
            if (w == 0) return false;
            p += w;
            n -= w;
        }
        return true;
    };
    bool ok = put(head.data(), head.size());
    for (size_t w = 0; ok && w < parts.size(); ++w) ok = (w == 0 || put(",", 1)) && put(parts[w].data(), parts[w].size());
    ok = ok && put("]}", 2);
    if (!ok) {
        const std::string m = std::strerror(errno);
        std::fclose(f);
        return Error::New("write " + path + ": " + m);
    }
    if (std::fclose(f) != 0) return Error::New("close " + path + ": " + std::strerror(errno));
    return Error::Nil();
}

namespace {

// the part of Runner.init that applies to a loaded or a fresh context alike
// (runner.go:111-114)
void stamp_workers(ExecutionContext& ec)
{
    for (size_t i = 0; i < ec.Workers.size(); ++i) {
        ec.Workers[i].id = (ec.StartWorkerID + (int64_t)i) % maxWorkerID;
        ec.Workers[i].minSize = ec.MinSize;
        ec.Workers[i].maxSize = ec.MaxSize;
    }
}

}  // namespace

std::pair<ExecutionContext, Error> LoadSavedContext(const char* json, size_t len)
{
    ExecutionContext ec;
    Reader r{json, json + len, json, std::string()};
    if (r.null()) {
        // json.Unmarshal of "null" into a struct pointer leaves the struct zero
    } else {
        const bool ok = read_object(r, [&](const std::string& k) {
            if (key_is(k, "endpoint")) return read_string_field(r, ec.Endpoint);
            if (key_is(k, "bucketNames")) {
                ec.BucketNames.clear();
                return read_array(r, [&] {
                    ec.BucketNames.emplace_back();
                    return read_string_field(r, ec.BucketNames.back());
                });
            }
            if (key_is(k, "numObj")) return read_int_field(r, ec.NumObj, "ExecutionContext.numObj");
            if (key_is(k, "numWorker")) return read_int_field(r, ec.NumWorker, "ExecutionContext.numWorker");
            if (key_is(k, "minSize")) return read_int_field(r, ec.MinSize, "ExecutionContext.minSize");
            if (key_is(k, "maxSize")) return read_int_field(r, ec.MaxSize, "ExecutionContext.maxSize");
            if (key_is(k, "startWorkerID")) return read_int_field(r, ec.StartWorkerID, "ExecutionContext.startWorkerID");
            if (key_is(k, "workers")) {
                ec.Workers.clear();
                if (read_workers_side_by_side(r, ec.Workers)) return true;
                return read_array(r, [&] {
                    ec.Workers.emplace_back();
                    return read_worker(r, ec.Workers.back());
                });
            }
            return r.skip();
        });
        if (!ok) return {ExecutionContext(), Error::New("json: " + r.err)};
    }
    r.ws();
    if (r.p != r.end) {
        r.fail("invalid character after top-level value");
        return {ExecutionContext(), Error::New("json: " + r.err)};
    }
    if (ec.BucketNames.empty()) return {ExecutionContext(), Error::New("bucket list is empty.")};   // runner.go:54-57
    stamp_workers(ec);
    std::vector<Error> tidy(ec.Workers.size(), Error::Nil());
    parallel_tasks(ec.Workers.size(), total_objects(ec), [&](size_t w) {
        for (BucketWithObject& b : ec.Workers[w].BucketsWithObject) {
            tidy[w] = b.ObjectMeta.TidyUp();
            if (!tidy[w].nil()) return;
        }
    });
    for (const Error& e : tidy)
        if (!e.nil()) return {ExecutionContext(), e};                // the first in table order
    return {std::move(ec), Error::Nil()};
}

std::pair<ExecutionContext, Error> LoadSavedContextFile(const std::string& path)
{
    const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return {ExecutionContext(), Error::New("open " + path + ": " + std::strerror(errno))};
    // A regular file is read where the page cache holds it (no copy of tens of
    // megabytes first); anything else (a pipe, an empty file) through read().
    struct stat st;
    if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void* m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        if (m != MAP_FAILED) {
            ::close(fd);
            auto out = LoadSavedContext(static_cast<const char*>(m), (size_t)st.st_size);
            ::munmap(m, (size_t)st.st_size);
            return out;
        }
    }
    std::string data;
    char buf[1 << 16];
    for (;;) {
        const ssize_t n = ::read(fd, buf, sizeof buf);
        if (n > 0) { data.append(buf, (size_t)n); continue; }
        if (n == 0) break;
        if (errno == EINTR) continue;
        const std::string m = std::strerror(errno);
        ::close(fd);
        return {ExecutionContext(), Error::New("read " + path + ": " + m)};
    }
    ::close(fd);
    return LoadSavedContext(data.data(), data.size());
}

std::pair<ExecutionContext, Error> NewExecutionContext(const std::string& endpoint,
                                                       const std::vector<std::string>& bucketNames,
                                                       int64_t numObj, int64_t numWorker, int64_t minSize,
                                                       int64_t maxSize, int64_t startWorkerID, int64_t runnerID)
{
    if (bucketNames.empty()) return {ExecutionContext(), Error::New("bucket list is empty.")};     // runner.go:54-57
    if (numWorker <= 0) return {ExecutionContext(), Error::New("numWorker must be positive")};
    if (numObj < 0) return {ExecutionContext(), Error::New("numObj must not be negative")};
    ExecutionContext ec;
    ec.Endpoint = endpoint;
    ec.BucketNames = bucketNames;
    ec.NumObj = numObj;
    ec.NumWorker = numWorker;
    ec.MinSize = minSize;
    ec.MaxSize = maxSize;
    ec.StartWorkerID = startWorkerID >= 0 ? startWorkerID : rand_intn(maxWorkerID);
    ec.Workers.resize((size_t)numWorker);
    stamp_workers(ec);
    // workers own disjoint key ranges (runner.go:117-125): their object lists are built side by side
    parallel_tasks((size_t)numWorker, (size_t)numObj * bucketNames.size(), [&](size_t i) {
        Worker& w = ec.Workers[i];
        w.BucketsWithObject.reserve(bucketNames.size());
        for (const std::string& name : bucketNames) {
            w.BucketsWithObject.emplace_back();
            w.BucketsWithObject.back().BucketName = name;
            w.BucketsWithObject.back().ObjectMeta =
                object::ObjectMeta(numObj / numWorker, (runnerID << 32) + ((int64_t)i << 24));
        }
    });
    return {std::move(ec), Error::Nil()};
}

uint64_t CountExisting(const ExecutionContext& ec)
{
    uint64_t n = 0;
    for (const Worker& w : ec.Workers)
        for (const BucketWithObject& b : w.BucketsWithObject) n += b.ObjectMeta.ExistingObjectIDs.size();
    return n;
}

Error PopulateExisting(ExecutionContext& ec, uint64_t seed, double fraction, int64_t maxWriteCount)
{
    if (maxWriteCount < 1) return Error::New("maxWriteCount must be at least 1");
    uint64_t state = seed;
    auto next = [&state]() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto unit = [&next]() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); };   // [0,1), 53 bits
    for (Worker& w : ec.Workers)
        for (BucketWithObject& b : w.BucketsWithObject) {
            object::ObjectMeta& om = b.ObjectMeta;
            for (size_t j = 0; j < om.ObjectList.size(); ++j) {
                if (!(unit() < fraction)) continue;
                auto size = pattern::DecideSizeFrom(ec.MinSize, ec.MaxSize, unit());
                if (!size.second.nil()) return size.second;
                object::Object& o = om.ObjectList[j];
                o.Size = size.first;
                o.WriteCount = 1 + (int64_t)(next() % (uint64_t)maxWriteCount);
                Error e = om.RegisterToExistingList(o.Key);
                if (!e.nil()) return e;
            }
        }
    return Error::Nil();
}

// One segment = the existing objects of one (worker, bucket) pair, in the order
// of its existingObjectIDs.  Segments are independent, so the table is built in
// two passes: sizes per segment (where each segment's descriptors and bodies
// start), then the descriptors themselves, both spread over a few host threads:
// for a million objects the table would otherwise take longer to build on one
// core (70 ms) than to validate on the device (1.3 ms) fifty times over.
namespace {

struct Segment {
    uint32_t worker = 0, bucket = 0;
    size_t first = 0;          // index of its first descriptor
    uint64_t units = 0;        // sum of its objects' units
    uint64_t offset = 0;       // where its first body starts
    Error err = Error::Nil();
};

template <class Fn>
void for_each_segment(std::vector<Segment>& segs, size_t n_objects, const Fn& fn)
{
    parallel_tasks(segs.size(), n_objects, [&](size_t i) { fn(segs[i]); });
}

}  // namespace

Error BuildExistingDescriptorsInto(const ExecutionContext& ec, uint64_t unixMicro, size_t workerBegin, size_t workerEnd,
                                   oval_obj_desc* descs, ObjectRef* refs, size_t cap, size_t* nOut, uint64_t* bodyBytesOut)
{
    if (workerEnd > ec.Workers.size()) workerEnd = ec.Workers.size();
    std::vector<Segment> segs;
    size_t n = 0;
    for (size_t w = workerBegin; w < workerEnd; ++w)
        for (size_t b = 0; b < ec.Workers[w].BucketsWithObject.size(); ++b) {
            Segment s;
            s.worker = (uint32_t)w;
            s.bucket = (uint32_t)b;
            s.first = n;
            n += ec.Workers[w].BucketsWithObject[b].ObjectMeta.ExistingObjectIDs.size();
            segs.push_back(std::move(s));
        }
    // pass 1: how many units each segment holds (and whether its IDs and sizes are sane)
    for_each_segment(segs, n, [&](Segment& s) {
        const object::ObjectMeta& om = ec.Workers[s.worker].BucketsWithObject[s.bucket].ObjectMeta;
        uint64_t units = 0;
        for (int64_t objID : om.ExistingObjectIDs) {
            if (objID < 0 || (uint64_t)objID >= om.ObjectList.size()) {
                s.err = Error::New("existingObjectIDs holds " + std::to_string(objID) + ", outside the object list");
                return;
            }
            const object::Object& obj = om.ObjectList[(size_t)objID];
            if (obj.Size < 0) { s.err = Error::New("object " + obj.Key + " has a negative size"); return; }
            const uint64_t u = (uint64_t)obj.Size / (uint64_t)pattern::dataUnitSize;
            if (u > UINT32_MAX) { s.err = Error::New("object " + obj.Key + " is too large"); return; }
            units += u;
        }
        s.units = units;
    });
    uint64_t off = 0;
    for (Segment& s : segs) {
        if (!s.err.nil()) return s.err;                          // the first in table order
        s.offset = off;
        off += s.units * (uint64_t)pattern::dataUnitSize;
    }
    if (nOut) *nOut = n;
    if (bodyBytesOut) *bodyBytesOut = off;
    if (!descs && !refs) return Error::Nil();                    // the caller only asked how large the table is
    // pass 2: the descriptors, bodies back to back in segment order; entries beyond `cap` are not written
    for_each_segment(segs, n, [&](Segment& s) {
        const Worker& wk = ec.Workers[s.worker];
        const BucketWithObject& bo = wk.BucketsWithObject[s.bucket];
        // Valid and Generate cut the bucket name to 16 bytes (pattern.go:74-76,119-121)
        const std::string bucket = bo.BucketName.substr(0, object::MaxBucketNameLength);
        uint64_t o = s.offset;
        size_t i = s.first;
        for (int64_t objID : bo.ObjectMeta.ExistingObjectIDs) {
            if (i >= cap) return;
            const object::Object& obj = bo.ObjectMeta.ObjectList[(size_t)objID];
            const uint64_t units = (uint64_t)obj.Size / (uint64_t)pattern::dataUnitSize;
            oval_obj_desc d;
            const int rc = oval_desc_init(&d, bucket.data(), bucket.size(), obj.Key.data(), obj.Key.size(),
                                          obj.WriteCount, wk.id, unixMicro, o, 0, (uint32_t)units);
            if (rc != OVAL_OK) {
                s.err = Error::New("object " + obj.Key + ": " + oval_last_error());   // this thread's message
                return;
            }
            if (descs) descs[i] = d;
            if (refs) refs[i] = ObjectRef{s.worker, s.bucket, objID};
            o += units * (uint64_t)pattern::dataUnitSize;
            ++i;
        }
    });
    for (const Segment& s : segs)
        if (!s.err.nil()) return s.err;
    return Error::Nil();
}

Error BuildExistingCompactInto(const ExecutionContext& ec, uint64_t unixMicro, size_t workerBegin, size_t workerEnd,
                               oval_obj_segment* segs, size_t segCap, oval_obj_rec* recs, size_t recCap,
                               size_t* nSegsOut, size_t* nObjOut, uint64_t* bodyBytesOut)
{
    if (workerEnd > ec.Workers.size()) workerEnd = ec.Workers.size();
    std::vector<Segment> parts;
    size_t n = 0;
    for (size_t w = workerBegin; w < workerEnd; ++w)
        for (size_t b = 0; b < ec.Workers[w].BucketsWithObject.size(); ++b) {
            const size_t have = ec.Workers[w].BucketsWithObject[b].ObjectMeta.ExistingObjectIDs.size();
            if (have == 0) continue;                             // a segment holds at least one object
            Segment s;
            s.worker = (uint32_t)w;
            s.bucket = (uint32_t)b;
            s.first = n;
            n += have;
            parts.push_back(std::move(s));
        }
    if (n >= 0xFFFFFFF0ull) return Error::New("too many existing objects for one batch");
    if (nSegsOut) *nSegsOut = parts.size();
    if (nObjOut) *nObjOut = n;
    const bool write = recs && segs && recCap >= n && segCap >= parts.size();
    for_each_segment(parts, n, [&](Segment& s) {
        const Worker& wk = ec.Workers[s.worker];
        const BucketWithObject& bo = wk.BucketsWithObject[s.bucket];
        const object::ObjectMeta& om = bo.ObjectMeta;
        if (om.KeyIDOffset < 0) { s.err = Error::New("negative keyIDOffset"); return; }
        uint64_t units = 0;
        size_t i = s.first;
        char key[24];
        for (int64_t objID : om.ExistingObjectIDs) {
            if (objID < 0 || (uint64_t)objID >= om.ObjectList.size() || objID > (int64_t)UINT32_MAX) {
                s.err = Error::New("existingObjectIDs holds " + std::to_string(objID) + ", outside the object list");
                return;
            }
            const object::Object& obj = om.ObjectList[(size_t)objID];
            if (obj.Size < 0) { s.err = Error::New("object " + obj.Key + " has a negative size"); return; }
            const uint64_t u = (uint64_t)obj.Size / (uint64_t)pattern::dataUnitSize;
            if (u > UINT32_MAX) { s.err = Error::New("object " + obj.Key + " is too large"); return; }
            if (!write) { units += u; continue; }                // sizes only: the keys are judged when the table is written
            // the key the device will write: "ov" + %010x (object.go:48-50)
            const uint64_t id = (uint64_t)om.KeyIDOffset + (uint64_t)objID;
            static const char hexd[] = "0123456789abcdef";
            key[0] = 'o'; key[1] = 'v';
            for (int d = 0; d < 10; ++d) key[2 + d] = hexd[(id >> (4 * (9 - d))) & 15u];
            if (id >> 40 || obj.Key.size() != 12 || std::memcmp(obj.Key.data(), key, 12) != 0) {
                s.err = Error::New("object " + obj.Key + " does not carry the key of ID " + std::to_string(id) +
                                   ": the compact form cannot describe it");
                return;
            }
            recs[i] = oval_obj_rec{(uint32_t)objID, (uint32_t)obj.WriteCount, (uint32_t)u};
            units += u;
            ++i;
        }
        s.units = units;
    });
    uint64_t off = 0;
    for (size_t k = 0; k < parts.size(); ++k) {
        Segment& s = parts[k];
        if (!s.err.nil()) return s.err;                          // the first in table order
        off += s.units * (uint64_t)pattern::dataUnitSize;
        if (!write) continue;
        const Worker& wk = ec.Workers[s.worker];
        const BucketWithObject& bo = wk.BucketsWithObject[s.bucket];
        // Valid and Generate cut the bucket name to 16 bytes (pattern.go:74-76,119-121)
        const std::string bucket = bo.BucketName.substr(0, object::MaxBucketNameLength);
        const int rc = oval_segment_init(&segs[k], bucket.data(), bucket.size(), wk.id, unixMicro,
                                         (uint64_t)bo.ObjectMeta.KeyIDOffset, (uint32_t)s.first,
                                         (uint32_t)bo.ObjectMeta.ExistingObjectIDs.size());
        if (rc != OVAL_OK) return Error::New("bucket " + bo.BucketName + ": " + oval_last_error());
    }
    if (bodyBytesOut) *bodyBytesOut = off;
    return Error::Nil();
}

std::pair<DescriptorTable, Error> BuildExistingDescriptors(const ExecutionContext& ec, uint64_t unixMicro,
                                                           size_t workerBegin, size_t workerEnd)
{
    DescriptorTable t;
    size_t n = 0;
    Error e = BuildExistingDescriptorsInto(ec, unixMicro, workerBegin, workerEnd, nullptr, nullptr, 0, &n, &t.bodyBytes);
    if (!e.nil()) return {DescriptorTable(), e};
    t.descs.resize(n);
    t.refs.resize(n);
    e = BuildExistingDescriptorsInto(ec, unixMicro, workerBegin, workerEnd, t.descs.data(), t.refs.data(), n, &n,
                                     &t.bodyBytes);
    if (!e.nil()) return {DescriptorTable(), e};
    return {std::move(t), Error::Nil()};
}

}  // namespace runner
}  // namespace oval

// ---------------------------------------------------------------------------
// C entry points (include/oval_context.h)
// ---------------------------------------------------------------------------
#include "../../include/oval_context.h"
#include "oval_capi_internal.h"

struct oval_execctx {
    oval::runner::ExecutionContext ec;
};
// An oval_objmeta* is an oval::object::ObjectMeta* (borrowed from a context, or
// owned when it came from oval_objmeta_new).
static inline oval::object::ObjectMeta* OM(oval_objmeta* p) { return reinterpret_cast<oval::object::ObjectMeta*>(p); }
static inline const oval::object::ObjectMeta* OM(const oval_objmeta* p)
{
    return reinterpret_cast<const oval::object::ObjectMeta*>(p);
}

namespace {

int put_err(const oval::Error& e, char* err, size_t cap)
{
    if (err && cap) err[0] = '\0';
    if (e.nil()) return 0;
    if (err && cap) {
        const size_t n = e.msg.size() < cap - 1 ? e.msg.size() : cap - 1;
        std::memcpy(err, e.msg.data(), n);
        err[n] = '\0';
    }
    return 1;
}

size_t put_text(const std::string& s, char* out, size_t cap)
{
    if (out && cap) std::memcpy(out, s.data(), s.size() < cap ? s.size() : cap);
    return s.size();
}

void put_key13(const std::string& k, char* out13)
{
    if (!out13) return;
    const size_t n = k.size() < 12 ? k.size() : 12;
    std::memcpy(out13, k.data(), n);
    out13[n] = '\0';
}

int64_t index_of(const oval::object::ObjectMeta* om, const oval::object::Object* o)
{
    return o ? (int64_t)(o - om->ObjectList.data()) : -1;
}

int adopt(std::pair<oval::runner::ExecutionContext, oval::Error>&& r, oval_execctx** out, char* err, size_t cap)
{
    if (out) *out = nullptr;
    if (!r.second.nil() || !out) return put_err(r.second.nil() ? oval::Error::New("out is NULL") : r.second, err, cap);
    oval_execctx* h = new (std::nothrow) oval_execctx;
    if (!h) return OVAL_E_NOMEM;
    h->ec = std::move(r.first);
    *out = h;
    return put_err(oval::Error::Nil(), err, cap);
}

}  // namespace

extern "C" {

int oval_execctx_new(const char* endpoint, const char* const* bucket_names, size_t n_buckets, int64_t num_obj,
                     int64_t num_worker, int64_t min_size, int64_t max_size, int64_t start_worker_id,
                     int64_t runner_id, oval_execctx** out, char* err, size_t err_cap)
try {
    std::vector<std::string> names;
    for (size_t i = 0; i < n_buckets; ++i) names.emplace_back(bucket_names && bucket_names[i] ? bucket_names[i] : "");
    return adopt(oval::runner::NewExecutionContext(endpoint ? endpoint : "", names, num_obj, num_worker, min_size,
                                                   max_size, start_worker_id, runner_id),
                 out, err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int oval_execctx_load(const char* json, size_t len, oval_execctx** out, char* err, size_t err_cap)
try {
    return adopt(oval::runner::LoadSavedContext(json ? json : "", json ? len : 0), out, err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int oval_execctx_load_file(const char* path, oval_execctx** out, char* err, size_t err_cap)
try {
    return adopt(oval::runner::LoadSavedContextFile(path ? path : ""), out, err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

size_t oval_execctx_marshal(const oval_execctx* ec, char* out, size_t cap)
try {
    if (!ec) return 0;
    return put_text(oval::runner::MarshalContext(ec->ec), out, cap);
} catch (...) {
    (void)oval_internal_caught();
    return 0;
}

int oval_execctx_save_file(const oval_execctx* ec, const char* path, char* err, size_t err_cap)
try {
    if (!ec || !path) return OVAL_E_INVALID;
    return put_err(oval::runner::SaveContext(ec->ec, path), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

void oval_execctx_destroy(oval_execctx* ec)
try { delete ec; } catch (...) {
    (void)oval_internal_caught();
}

void oval_execctx_info(const oval_execctx* ec, int64_t* num_obj, int64_t* num_worker, int64_t* min_size,
                       int64_t* max_size, int64_t* start_worker_id, size_t* n_workers, size_t* n_bucket_names)
try {
    if (!ec) return;
    if (num_obj) *num_obj = ec->ec.NumObj;
    if (num_worker) *num_worker = ec->ec.NumWorker;
    if (min_size) *min_size = ec->ec.MinSize;
    if (max_size) *max_size = ec->ec.MaxSize;
    if (start_worker_id) *start_worker_id = ec->ec.StartWorkerID;
    if (n_workers) *n_workers = ec->ec.Workers.size();
    if (n_bucket_names) *n_bucket_names = ec->ec.BucketNames.size();
} catch (...) {
    (void)oval_internal_caught();
}

size_t oval_execctx_endpoint(const oval_execctx* ec, char* out, size_t cap)
try {
    return ec ? put_text(ec->ec.Endpoint, out, cap) : 0;
} catch (...) {
    (void)oval_internal_caught();
    return 0;
}

int64_t oval_execctx_worker_id(const oval_execctx* ec, size_t worker)
try {
    return ec && worker < ec->ec.Workers.size() ? ec->ec.Workers[worker].id : -1;
} catch (...) {
    (void)oval_internal_caught();
    return -1;
}

size_t oval_execctx_worker_buckets(const oval_execctx* ec, size_t worker)
try {
    return ec && worker < ec->ec.Workers.size() ? ec->ec.Workers[worker].BucketsWithObject.size() : 0;
} catch (...) {
    (void)oval_internal_caught();
    return 0;
}

size_t oval_execctx_bucket_name(const oval_execctx* ec, size_t worker, size_t bucket, char* out, size_t cap)
try {
    if (!ec || worker >= ec->ec.Workers.size() || bucket >= ec->ec.Workers[worker].BucketsWithObject.size()) return 0;
    return put_text(ec->ec.Workers[worker].BucketsWithObject[bucket].BucketName, out, cap);
} catch (...) {
    (void)oval_internal_caught();
    return 0;
}

oval_objmeta* oval_execctx_objmeta(oval_execctx* ec, size_t worker, size_t bucket)
try {
    if (!ec || worker >= ec->ec.Workers.size() || bucket >= ec->ec.Workers[worker].BucketsWithObject.size())
        return nullptr;
    return reinterpret_cast<oval_objmeta*>(&ec->ec.Workers[worker].BucketsWithObject[bucket].ObjectMeta);
} catch (...) {
    (void)oval_internal_caught();
    return nullptr;
}

uint64_t oval_execctx_count_existing(const oval_execctx* ec)
try { return ec ? oval::runner::CountExisting(ec->ec) : 0; } catch (...) {
    (void)oval_internal_caught();
    return 0;
}

oval_objmeta* oval_objmeta_new(int64_t num_obj, int64_t key_id_offset)
try {
    return reinterpret_cast<oval_objmeta*>(new (std::nothrow) oval::object::ObjectMeta(num_obj, key_id_offset));
} catch (...) {
    (void)oval_internal_caught();
    return nullptr;
}

void oval_objmeta_free(oval_objmeta* om)
try { delete OM(om); } catch (...) {
    (void)oval_internal_caught();
}

int64_t oval_objmeta_get_random_object(oval_objmeta* om)
try { return om ? index_of(OM(om), OM(om)->GetRandomObject()) : -1; } catch (...) {
    (void)oval_internal_caught();
    return -1;
}

int oval_objmeta_register_to_existing_list(oval_objmeta* om, const char* key, size_t key_len, char* err,
                                           size_t err_cap)
try {
    if (!om) return OVAL_E_INVALID;
    return put_err(OM(om)->RegisterToExistingList(std::string(key ? key : "", key ? key_len : 0)), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int64_t oval_objmeta_pop_existing_random_object(oval_objmeta* om)
try {
    return om ? index_of(OM(om), OM(om)->PopExistingRandomObject()) : -1;
} catch (...) {
    (void)oval_internal_caught();
    return -1;
}

int64_t oval_objmeta_get_existing_random_object(oval_objmeta* om)
try {
    return om ? index_of(OM(om), OM(om)->GetExistingRandomObject()) : -1;
} catch (...) {
    (void)oval_internal_caught();
    return -1;
}

int oval_objmeta_exist(const oval_objmeta* om, const char* key, size_t key_len, int* exists_out, char* err,
                       size_t err_cap)
try {
    if (!om) return OVAL_E_INVALID;
    auto r = OM(om)->Exist(std::string(key ? key : "", key ? key_len : 0));
    if (exists_out) *exists_out = r.first ? 1 : 0;
    return put_err(r.second, err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

void oval_objmeta_get_head_and_tail_key(const oval_objmeta* om, char* head13, char* tail13)
try {
    if (!om) return;
    auto r = OM(om)->GetHeadAndTailKey();
    put_key13(r.first, head13);
    put_key13(r.second, tail13);
} catch (...) {
    (void)oval_internal_caught();
}

int oval_objmeta_tidy_up(oval_objmeta* om, char* err, size_t err_cap)
try {
    if (!om) return OVAL_E_INVALID;
    return put_err(OM(om)->TidyUp(), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

size_t oval_objmeta_num_objects(const oval_objmeta* om)
try { return om ? OM(om)->ObjectList.size() : 0; } catch (...) {
    (void)oval_internal_caught();
    return 0;
}
size_t oval_objmeta_num_existing(const oval_objmeta* om)
try { return om ? OM(om)->ExistingObjectIDs.size() : 0; } catch (...) {
    (void)oval_internal_caught();
    return 0;
}

int64_t oval_objmeta_existing_id(const oval_objmeta* om, size_t i)
try {
    return om && i < OM(om)->ExistingObjectIDs.size() ? OM(om)->ExistingObjectIDs[i] : -1;
} catch (...) {
    (void)oval_internal_caught();
    return -1;
}

int64_t oval_objmeta_key_id_offset(const oval_objmeta* om)
try { return om ? OM(om)->KeyIDOffset : 0; } catch (...) {
    (void)oval_internal_caught();
    return -1;
}

size_t oval_objmeta_key_prefix(const oval_objmeta* om, char* out, size_t cap)
try {
    return om ? put_text(OM(om)->KeyPrefix, out, cap) : 0;
} catch (...) {
    (void)oval_internal_caught();
    return 0;
}

int oval_objmeta_get_object(const oval_objmeta* om, int64_t index, char* key13, int64_t* size, int64_t* write_count)
try {
    if (!om) return OVAL_E_INVALID;
    if (index < 0 || (uint64_t)index >= OM(om)->ObjectList.size()) return OVAL_E_RANGE;
    const oval::object::Object& o = OM(om)->ObjectList[(size_t)index];
    put_key13(o.Key, key13);
    if (size) *size = o.Size;
    if (write_count) *write_count = o.WriteCount;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

int oval_objmeta_set_object(oval_objmeta* om, int64_t index, int64_t size, int64_t write_count)
try {
    if (!om) return OVAL_E_INVALID;
    if (index < 0 || (uint64_t)index >= OM(om)->ObjectList.size()) return OVAL_E_RANGE;
    OM(om)->ObjectList[(size_t)index].Size = size;
    OM(om)->ObjectList[(size_t)index].WriteCount = write_count;
    return OVAL_OK;
} catch (...) {
    return oval_internal_caught();
}

void oval_object_seed_random(uint64_t seed)
try { oval::object::SeedRandom(seed); } catch (...) {
    (void)oval_internal_caught();
}

int oval_execctx_populate(oval_execctx* ec, uint64_t seed, double fraction, int64_t max_write_count, char* err,
                          size_t err_cap)
try {
    if (!ec) return OVAL_E_INVALID;
    return put_err(oval::runner::PopulateExisting(ec->ec, seed, fraction, max_write_count), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int oval_execctx_build_descs(const oval_execctx* ec, uint64_t unix_micro, size_t worker_begin, size_t worker_end,
                             oval_obj_desc* descs_out, oval_obj_ref* refs_out, uint64_t cap, uint64_t* n_out,
                             uint64_t* body_bytes_out, char* err, size_t err_cap)
try {
    if (!ec) return OVAL_E_INVALID;
    static_assert(sizeof(oval_obj_ref) == sizeof(oval::runner::ObjectRef) && offsetof(oval_obj_ref, obj_id) == 8,
                  "oval_obj_ref is ObjectRef");
    size_t n = 0;
    uint64_t bytes = 0;
    const oval::Error e = oval::runner::BuildExistingDescriptorsInto(
        ec->ec, unix_micro, worker_begin, worker_end, descs_out, reinterpret_cast<oval::runner::ObjectRef*>(refs_out),
        (size_t)cap, &n, &bytes);
    if (!e.nil()) return put_err(e, err, err_cap);
    if (n_out) *n_out = n;
    if (body_bytes_out) *body_bytes_out = bytes;
    return put_err(oval::Error::Nil(), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

int oval_execctx_build_compact(const oval_execctx* ec, uint64_t unix_micro, size_t worker_begin, size_t worker_end,
                               oval_obj_segment* segs_out, uint64_t seg_cap, oval_obj_rec* recs_out, uint64_t rec_cap,
                               uint64_t* n_segs_out, uint64_t* n_obj_out, uint64_t* body_bytes_out, char* err,
                               size_t err_cap)
try {
    if (!ec) return OVAL_E_INVALID;
    size_t n_segs = 0, n_obj = 0;
    uint64_t bytes = 0;
    const oval::Error e = oval::runner::BuildExistingCompactInto(ec->ec, unix_micro, worker_begin, worker_end, segs_out,
                                                                 (size_t)seg_cap, recs_out, (size_t)rec_cap, &n_segs,
                                                                 &n_obj, &bytes);
    if (!e.nil()) return put_err(e, err, err_cap);
    if (n_segs_out) *n_segs_out = n_segs;
    if (n_obj_out) *n_obj_out = n_obj;
    if (body_bytes_out) *body_bytes_out = bytes;
    return put_err(oval::Error::Nil(), err, err_cap);
} catch (...) {
    return oval_internal_caught_text(err, err_cap);
}

}  // extern "C"
