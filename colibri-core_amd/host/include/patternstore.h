// patternstore.h — host-side result containers of the C++ face.
//   IndexedCorpus : the whole .colibri.dat payload in memory + sentence offsets (reference include/patternstore.h:43-401,
//                   src/pattern.cpp:1900-2162). The same "flat byte array + per-sentence offset vector" is what the
//                   device keeps in HBM; here it serves callers that hand a preloaded corpus to a model.
//   PatternMap<V> : std::unordered_map<Pattern,V> with the reference's accessors (include/patternstore.h:937-1011).
//   PatternSet<>  : std::unordered_set<Pattern> (only needed as the `filter` argument type of train()).
#ifndef COLIBRI_AMD_PATTERNSTORE_H
#define COLIBRI_AMD_PATTERNSTORE_H
#include <cstdint>
#include <functional>
#include <istream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "classdecoder.h"
#include "common.h"
#include "datatypes.h"
#include "pattern.h"

class IndexedCorpus {
  public:
    IndexedCorpus() : corpus(NULL), corpussize(0) {}
    explicit IndexedCorpus(std::istream& in, bool debug = false) : corpus(NULL), corpussize(0) { load(in, debug); }
    explicit IndexedCorpus(const std::string& filename, bool debug = false) : corpus(NULL), corpussize(0) { load(filename, debug); }
    ~IndexedCorpus() { delete[] corpus; }
    IndexedCorpus(const IndexedCorpus&)            = delete;
    IndexedCorpus& operator=(const IndexedCorpus&) = delete;

    void load(std::istream& in, bool debug = false);
    void load(const std::string& filename, bool debug = false);

    unsigned char* beginpointer() const { return corpus; }
    size_t         bytesize() const { return corpussize; }
    /** number of sentences; every delimiter opens a new one, empty ones included (src/pattern.cpp:1947-1958) */
    unsigned int sentences() const { return (unsigned int)sentencestart.size(); }
    bool         empty() const { return corpussize == 0; }
    /** total tokens */
    size_t size() const { return colibri_host::token_count(corpus, corpussize) - delimiters; }
    /** 1-based sentence as a pointer into the corpus (delimiter excluded) */
    PatternPointer getsentence(int sentence) const;
    unsigned int   sentencelength(int sentence) const { return (unsigned int)getsentence(sentence).n(); }
    /** `length` tokens starting at a corpus position; throws KeyError outside the corpus */
    PatternPointer getpattern(const IndexReference& begin, int length = 1) const;

  private:
    unsigned char*        corpus;
    size_t                corpussize;
    size_t                delimiters = 0;
    std::vector<uint64_t> sentencestart;  // byte offset of each sentence
};

template <class ValueType>
class PatternMap {
  public:
    typedef std::unordered_map<Pattern, ValueType>    ContainerType;
    typedef typename ContainerType::iterator          iterator;
    typedef typename ContainerType::const_iterator    const_iterator;
    ContainerType data;

    virtual ~PatternMap() {}
    void   insert(const Pattern& p, const ValueType& v) { materialise(); data[p] = v; }
    void   insert(const Pattern& p) { materialise(); data[p]; }
    bool   has(const Pattern& p) const { materialise(); return data.find(p) != data.end(); }
    bool   has(const PatternPointer& p) const { return has(Pattern(p)); }
    size_t size() const { return pending_size ? pending_size() : data.size(); }
    void   reserve(size_t s) { data.reserve(s); }
    ValueType& operator[](const Pattern& p) { materialise(); return data[p]; }
    ValueType& operator[](const PatternPointer& p) { materialise(); return data[Pattern(p)]; }
    iterator       begin() { materialise(); return data.begin(); }
    const_iterator begin() const { materialise(); return data.begin(); }
    iterator       end() { materialise(); return data.end(); }
    const_iterator end() const { materialise(); return data.end(); }
    iterator       find(const Pattern& p) { materialise(); return data.find(p); }
    const_iterator find(const Pattern& p) const { materialise(); return data.find(p); }
    iterator       find(const PatternPointer& p) { return find(Pattern(p)); }
    bool           erase(const Pattern& p) { materialise(); return data.erase(p) > 0; }
    iterator       erase(const_iterator pos) { materialise(); return data.erase(pos); }

  protected:
    // Results of a device training run stay in flat arrays until a caller actually walks the map
    // (train -> write never builds ten million heap Patterns). The model installs these two hooks.
    mutable std::function<void()>   pending_fill;
    mutable std::function<size_t()> pending_size;
    void materialise() const {
        if (pending_fill) {
            std::function<void()> f;
            f.swap(pending_fill);
            pending_size = nullptr;
            f();
        }
    }
};

template <class ValueType = uint32_t>
class PatternSet {
  public:
    std::unordered_set<Pattern> data;
    typedef typename std::unordered_set<Pattern>::iterator iterator;
    void     insert(const Pattern& p) { data.insert(p); }
    bool     has(const Pattern& p) const { return data.count(p) != 0; }
    bool     has(const PatternPointer& p) const { return data.count(Pattern(p)) != 0; }
    size_t   size() const { return data.size(); }
    iterator begin() { return data.begin(); }
    iterator end() { return data.end(); }
};
#endif
