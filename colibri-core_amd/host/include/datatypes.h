// datatypes.h — value types of the models: IndexReference, IndexedData and their (de)serialisers.
// reference: include/datatypes.h:33-89 (IndexReference), :95-180 (IndexedData), :210-239 (BaseValueHandler),
// :247-297 (IndexedDataHandler). On disk: u32 count, then count x (u32 sentence, u16 token) (:263-270, :55-58).
#ifndef COLIBRI_AMD_DATATYPES_H
#define COLIBRI_AMD_DATATYPES_H
#include <algorithm>
#include <cstdint>
#include <iostream>
#include <string>
#include <vector>

class IndexReference {
  public:
    uint32_t sentence;  ///< 1-based, empty sentences are numbered too
    uint16_t token;     ///< 0-based
    IndexReference() : sentence(0), token(0) {}
    explicit IndexReference(uint32_t s, uint16_t t) : sentence(s), token(t) {}
    explicit IndexReference(std::istream& in) {
        in.read((char*)&sentence, sizeof(uint32_t));
        in.read((char*)&token, sizeof(uint16_t));
    }
    void write(std::ostream& out) const {
        out.write((const char*)&sentence, sizeof(uint32_t));
        out.write((const char*)&token, sizeof(uint16_t));
    }
    bool operator<(const IndexReference& o) const { return sentence < o.sentence || (sentence == o.sentence && token < o.token); }
    bool operator>(const IndexReference& o) const { return o < *this; }
    bool operator==(const IndexReference& o) const { return sentence == o.sentence && token == o.token; }
    bool operator!=(const IndexReference& o) const { return !(*this == o); }
    IndexReference operator+(const int d) const { return IndexReference(sentence, (uint16_t)(token + d)); }
    std::string tostring() const { return std::to_string(sentence) + ":" + std::to_string(token); }
    friend std::ostream& operator<<(std::ostream& out, const IndexReference& r) { return out << r.tostring(); }
};

class IndexedData {
  public:
    std::vector<IndexReference> data;
    IndexedData() {}
    typedef std::vector<IndexReference>::iterator       iterator;
    typedef std::vector<IndexReference>::const_iterator const_iterator;
    iterator       begin() { return data.begin(); }
    const_iterator begin() const { return data.begin(); }
    iterator       end() { return data.end(); }
    const_iterator end() const { return data.end(); }
    unsigned int   count() const { return (unsigned int)data.size(); }
    size_t         size() const { return data.size(); }
    void           insert(IndexReference ref) { data.push_back(ref); }
    void           sort() { std::sort(data.begin(), data.end()); }
    bool           has(const IndexReference& ref, bool sorted = false) const {
        return sorted ? std::binary_search(data.begin(), data.end(), ref) : std::find(data.begin(), data.end(), ref) != data.end();
    }
    bool operator==(const IndexedData& o) const { return data == o.data; }
};

/** value handlers: how a model's value is read, written, counted and incremented */
template <class ValueType>
class BaseValueHandler {
  public:
    void         read(std::istream& in, ValueType& v) { in.read((char*)&v, sizeof(ValueType)); }
    void         write(std::ostream& out, const ValueType& v) { out.write((const char*)&v, sizeof(ValueType)); }
    unsigned int count(const ValueType& v) const { return (unsigned int)v; }
    void         add(ValueType* v, const IndexReference&) const { *v = *v + 1; }
};

class IndexedDataHandler {
  public:
    void read(std::istream& in, IndexedData& v) {
        uint32_t c = 0;
        in.read((char*)&c, sizeof(uint32_t));
        v.data.reserve(c);
        for (uint32_t i = 0; i < c; ++i) v.data.push_back(IndexReference(in));
    }
    void write(std::ostream& out, const IndexedData& v) {
        const uint32_t c = v.count();
        out.write((const char*)&c, sizeof(uint32_t));
        for (const IndexReference& r : v.data) r.write(out);
    }
    unsigned int count(const IndexedData& v) const { return v.count(); }
    void         add(IndexedData* v, const IndexReference& ref) const { v->insert(ref); }
};
#endif
