// pattern.h — Pattern / PatternPointer value types of the C++ face.
//
// Same names, members and meaning as the reference's key types (include/pattern.h:73-354 Pattern,
// :361-552 PatternPointer), written from scratch and only as far as the hot path's callers need:
// construction from bytes / corpus slices, n(), bytesize(), category(), hash() (the same 64-bit
// SpookyHash the device computes), equality, ordering, (de)serialisation in the .colibri.patternmodel
// key format (key bytes + 00, reference src/pattern.cpp:268-277, :483-587) and decoding to text.
#ifndef COLIBRI_AMD_PATTERN_H
#define COLIBRI_AMD_PATTERN_H
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

class ClassDecoder;
class PatternPointer;

namespace colibri_host {
uint64_t spooky_hash64(const unsigned char* data, size_t len);  // SpookyHash::Hash64 (reference include/SpookyV2.h:59-66)
size_t   token_count(const unsigned char* data, size_t bytes);  // bytes < 128 end a token (reference src/pattern.cpp:74-105)
size_t   key_bytesize(const unsigned char* data);               // up to the first 00 that does not follow a high byte (:59-72)
PatternCategory category_of(const unsigned char* data, size_t bytes);
std::string decode_key(const unsigned char* data, size_t bytes, const ClassDecoder& decoder);
}  // namespace colibri_host

/** Owning pattern: heap bytes terminated by 00 (reference include/pattern.h:73-354). */
class Pattern {
  public:
    unsigned char* data;  ///< key bytes + terminating 00, or NULL for the empty pattern

    Pattern() : data(NULL) {}
    Pattern(const unsigned char* bytes, size_t size) { assign(bytes, size); }
    Pattern(const Pattern& o) { assign(o.data, o.data ? o.bytesize() : 0); }
    Pattern(Pattern&& o) noexcept : data(o.data) { o.data = NULL; }
    Pattern(const PatternPointer& pp);  // materialises a masked pointer: gapped tokens become the byte 03 (src/pattern.cpp:873-909)
    /** reads one pattern (key bytes up to the unescaped 00) from a model / corpus stream, v2 encoding (src/pattern.cpp:483-587) */
    explicit Pattern(std::istream& in, bool ignoreeol = false, const unsigned char version = 2);
    ~Pattern() { delete[] data; }
    Pattern& operator=(const Pattern& o) {
        if (this != &o) {
            delete[] data;
            assign(o.data, o.data ? o.bytesize() : 0);
        }
        return *this;
    }
    Pattern& operator=(Pattern&& o) noexcept {
        std::swap(data, o.data);
        return *this;
    }

    size_t bytesize() const { return data ? colibri_host::key_bytesize(data) : 0; }
    size_t n() const { return data ? colibri_host::token_count(data, bytesize()) : 0; }
    size_t size() const { return n(); }
    size_t hash() const { return (data == NULL || data[0] == 0) ? 0 : (size_t)colibri_host::spooky_hash64(data, bytesize()); }  // src/pattern.cpp:234-238
    PatternCategory category() const { return colibri_host::category_of(data, bytesize()); }
    bool isskipgram() const { return category() == SKIPGRAM; }
    bool isflexgram() const { return category() == FLEXGRAM; }

    bool operator==(const Pattern& o) const {
        const size_t a = bytesize(), b = o.bytesize();
        return a == b && (a == 0 || std::memcmp(data, o.data, a) == 0);
    }
    bool operator!=(const Pattern& o) const { return !(*this == o); }
    bool operator<(const Pattern& o) const {
        const size_t a = bytesize(), b = o.bytesize();
        const int    c = std::memcmp(data ? data : (const unsigned char*)"", o.data ? o.data : (const unsigned char*)"", a < b ? a : b);
        return c != 0 ? c < 0 : a < b;
    }

    void write(std::ostream& out, const unsigned char* corpusstart = NULL) const;  // key bytes + 00 (src/pattern.cpp:268-277)
    std::string tostring(const ClassDecoder& decoder) const;
    std::string tohex() const;

    /** all windows of n tokens, left to right (skipgram windows included: a gap is a token); returns how many were added
     *  (reference include/pattern.h:251-273, src/pattern.cpp:1216-1282). The pair forms carry the token offset of the window. */
    int ngrams(std::vector<Pattern>& container, const int n) const;
    int ngrams(std::vector<PatternPointer>& container, const int n) const;
    int ngrams(std::vector<std::pair<Pattern, int>>& container, const int n) const;
    int ngrams(std::vector<std::pair<PatternPointer, int>>& container, const int n) const;
    /** the windows of every size minn..maxn (maxn clipped to the pattern's length), size by size (include/pattern.h:259-282, src/pattern.cpp:1298-1374) */
    int subngrams(std::vector<Pattern>& container, int minn = 1, int maxn = 99) const;
    int subngrams(std::vector<PatternPointer>& container, int minn = 1, int maxn = 99) const;
    int subngrams(std::vector<std::pair<Pattern, int>>& container, int minn = 1, int maxn = 9) const;
    int subngrams(std::vector<std::pair<PatternPointer, int>>& container, int minn = 1, int maxn = 9) const;

  private:
    void assign(const unsigned char* bytes, size_t size) {
        if (bytes == NULL || size == 0) {
            data = NULL;
            return;
        }
        data = new unsigned char[size + 1];
        std::memcpy(data, bytes, size);
        data[size] = 0;
    }
};

/** Non-owning view {data, bytes, mask} into corpus bytes (reference include/pattern.h:361-552). */
class PatternPointer {
  public:
    unsigned char* data;
    uint64_t       bytes;
    uint32_t       mask;  ///< bit i set = token i is a gap; bit 31 = flexgram

    PatternPointer() : data(NULL), bytes(0), mask(0) {}
    PatternPointer(unsigned char* d, uint64_t b, uint32_t m = 0) : data(d), bytes(b), mask(m) {}
    explicit PatternPointer(const Pattern* p) : data(p->data), bytes(p->bytesize()), mask(0) {}
    explicit PatternPointer(const Pattern& p) : data(p.data), bytes(p.bytesize()), mask(0) {}

    size_t bytesize() const { return bytes; }
    size_t n() const { return colibri_host::token_count(data, bytes); }
    size_t size() const { return n(); }
    bool   isgap(int index) const { return mask != 0 && index <= 30 && (mask & bitmask_of(index)); }
    bool   isflexgram() const { return (mask >> 31) != 0; }
    PatternCategory category() const { return mask == 0 ? colibri_host::category_of(data, bytes) : (isflexgram() ? FLEXGRAM : SKIPGRAM); }
    size_t hash() const { return Pattern(*this).hash(); }
    /** all n-token windows with their token offset (reference src/pattern.cpp:1284-1296) */
    int ngrams(std::vector<std::pair<PatternPointer, int>>& container, const int n) const;
    int ngrams(std::vector<PatternPointer>& container, const int n) const;
    /** the windows of every size minn..maxn, size by size (reference include/pattern.h:503-506, src/pattern.cpp:1324-1374) */
    int subngrams(std::vector<PatternPointer>& container, int minn = 1, int maxn = 9) const;
    int subngrams(std::vector<std::pair<PatternPointer, int>>& container, int minn = 1, int maxn = 9) const;
    bool operator==(const PatternPointer& o) const { return Pattern(*this) == Pattern(o); }
    std::string tostring(const ClassDecoder& decoder) const { return Pattern(*this).tostring(decoder); }
};

namespace std {
template <>
struct hash<Pattern> {
    size_t operator()(const Pattern& p) const { return p.hash(); }
};
template <>
struct hash<PatternPointer> {
    size_t operator()(const PatternPointer& p) const { return p.hash(); }
};
}  // namespace std
#endif
