// patternmodel.h — PatternModelOptions / PatternModel<uint32_t> / IndexedPatternModel<> : the C++ face of the
// MI355X-native pattern-model builder. Same class names, method names, argument orders and error behaviour as the
// reference (include/patternmodel.h:103-213 options, :546-2674 PatternModel, :2682-3875 IndexedPatternModel) so that
// callers such as src/patternmodeller.cpp:316-319 or src/benchmarks.cpp:228-237 compile against it unchanged; what
// train() does is different: the corpus goes to HBM and the whole informed-iterative counting (reference :880-1345)
// runs in libcolibri_hip.so through the C ABI of include/colibri_hip.h. There is no host implementation of the
// counting loop: options outside the accelerated subset raise InternalError after a message on stderr, exactly as the
// reference reports its own errors (include/common.h:41-44).
#ifndef COLIBRI_AMD_PATTERNMODEL_H
#define COLIBRI_AMD_PATTERNMODEL_H
#include <cstdint>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "colibri_hip.h"
#include "patternstore.h"

enum ModelType {
    UNINDEXEDPATTERNMODEL        = 10,
    UNINDEXEDPATTERNPOINTERMODEL = 11,
    INDEXEDPATTERNMODEL          = 20,
    INDEXEDPATTERNPOINTERMODEL   = 21,
    PATTERNSETMODEL              = 30,
    PATTERNALIGNMENTMODEL        = 40,
};

/** reads the type byte of a model file (reference src/patternmodel.cpp:3-15) */
int getmodeltype(const std::string& filename);

class NoSuchPattern : public std::exception {
    const char* what() const throw() override { return "Pattern not found in model"; }
};

/** Options for training / loading — public fields and defaults of reference include/patternmodel.h:103-180. */
class PatternModelOptions {
  public:
    int  MINTOKENS, MINTOKENS_SKIPGRAMS, MINTOKENS_UNIGRAMS, MINLENGTH, MAXLENGTH, MAXBACKOFFLENGTH;
    bool DOSKIPGRAMS, DOSKIPGRAMS_EXHAUSTIVE;
    int  MINSKIPTYPES, MAXSKIPS;
    bool DOREVERSEINDEX, DOPATTERNPERLINE;
    int  PRUNENONSUBSUMED, PRUNESUBSUMED;
    bool DOREMOVEINDEX, DOREMOVENGRAMS, DOREMOVESKIPGRAMS, DOREMOVEFLEXGRAMS, DORESET;
    bool QUIET, DEBUG;
    PatternModelOptions()
        : MINTOKENS(-1), MINTOKENS_SKIPGRAMS(-1), MINTOKENS_UNIGRAMS(1), MINLENGTH(1), MAXLENGTH(100), MAXBACKOFFLENGTH(100), DOSKIPGRAMS(false),
          DOSKIPGRAMS_EXHAUSTIVE(false), MINSKIPTYPES(2), MAXSKIPS(3), DOREVERSEINDEX(true), DOPATTERNPERLINE(false), PRUNENONSUBSUMED(0), PRUNESUBSUMED(0),
          DOREMOVEINDEX(false), DOREMOVENGRAMS(false), DOREMOVESKIPGRAMS(false), DOREMOVEFLEXGRAMS(false), DORESET(false), QUIET(false), DEBUG(false) {}
};

/** The limited polymorphic view other components take of a model (reference include/patternmodel.h:234-287). */
class PatternModelInterface {
  public:
    virtual ~PatternModelInterface() {}
    virtual int          getmodeltype() const                       = 0;
    virtual int          getmodelversion() const                    = 0;
    virtual bool         has(const Pattern&) const                  = 0;
    virtual bool         has(const PatternPointer&) const           = 0;
    virtual size_t       size() const                               = 0;
    virtual unsigned int occurrencecount(const Pattern& pattern)    = 0;
    virtual double       frequency(const Pattern&)                  = 0;
    virtual int          maxlength() const                          = 0;
    virtual int          minlength() const                          = 0;
    virtual unsigned int types()                                    = 0;
    virtual unsigned int tokens() const                             = 0;
};

namespace colibri_host {

/** flat result of one device training run, as exported through colibri_export_* */
struct TrainResult {
    colibri_stats              stats{};
    std::vector<uint64_t>      key_off;
    std::vector<unsigned char> key_bytes;
    std::vector<uint32_t>      counts;
    std::vector<uint64_t>      ref_off;
    std::vector<uint32_t>      ref_sentence;
    std::vector<uint16_t>      ref_token;
    size_t                     size() const { return counts.size(); }
};

/** whole corpus file -> v2 payload (header stripped; v1 data converted, reference src/classencoder.cpp:602-647) */
std::vector<unsigned char> read_corpus_payload(std::istream& in);
/** upload + train + export through the C ABI; prints the library's message on stderr and throws InternalError on any status != 0 */
void device_train(const unsigned char* payload, uint64_t nbytes, const colibri_options& opt, uint32_t firstsentence, TrainResult& out);
/** the per-order progress lines the reference prints while training (patternmodel.h:1005-1019, :1195-1245) */
void print_training_log(const colibri_stats& s, const colibri_options& o, std::ostream& err);
/** reads one pattern of a model file in the given class-encoding version (1 or 2) and returns it v2-encoded */
Pattern read_model_pattern(std::istream& in, unsigned char classencodingversion);

inline void value_from_result(const TrainResult& r, size_t j, uint32_t& v) { v = r.counts[j]; }
inline void value_from_result(const TrainResult& r, size_t j, IndexedData& v) {
    v.data.clear();
    if (r.ref_off.empty()) return;
    for (uint64_t k = r.ref_off[j]; k < r.ref_off[j + 1]; ++k) v.data.push_back(IndexReference(r.ref_sentence[k], r.ref_token[k]));
}
inline void write_value_from_result(std::ostream& out, const TrainResult& r, size_t j, const uint32_t*) { out.write((const char*)&r.counts[j], sizeof(uint32_t)); }
inline void write_value_from_result(std::ostream& out, const TrainResult& r, size_t j, const IndexedData*) {
    const uint32_t c = r.ref_off.empty() ? 0 : (uint32_t)(r.ref_off[j + 1] - r.ref_off[j]);
    out.write((const char*)&c, sizeof(uint32_t));
    for (uint32_t k = 0; k < c; ++k) IndexReference(r.ref_sentence[r.ref_off[j] + k], r.ref_token[r.ref_off[j] + k]).write(out);
}
template <class V>
struct is_indexed_value {
    static const bool value = false;
};
template <>
struct is_indexed_value<IndexedData> {
    static const bool value = true;
};

}  // namespace colibri_host

/**
 * A pattern model: pattern -> value (count, or index of occurrences). reference include/patternmodel.h:546-2674.
 */
template <class ValueType, class ValueHandler = BaseValueHandler<ValueType>, class MapType = PatternMap<ValueType>>
class PatternModel : public MapType, public PatternModelInterface {
  protected:
    unsigned char model_type, model_version;
    uint64_t      totaltokens, totaltypes;
    int           maxn, minn;
    bool          hasskipgrams_;
    ValueHandler  valuehandler;
    std::shared_ptr<colibri_host::TrainResult> result;  // device results not yet turned into map nodes

    void install_result(std::shared_ptr<colibri_host::TrainResult> r) {
        result = r;
        this->data.clear();
        PatternModel* self = this;
        this->pending_size = [r]() { return r->size(); };
        this->pending_fill = [self, r]() {
            self->data.reserve(r->size());
            for (size_t j = 0; j < r->size(); ++j) {
                ValueType v{};
                colibri_host::value_from_result(*r, j, v);
                self->data.emplace(Pattern(r->key_bytes.data() + r->key_off[j], (size_t)(r->key_off[j + 1] - r->key_off[j])), std::move(v));
            }
            self->result.reset();
        };
    }

  public:
    IndexedCorpus* reverseindex;
    bool           reverseindex_internal;

    /** empty model, optionally attached to a preloaded corpus (reference :644) */
    PatternModel<ValueType, ValueHandler, MapType>(IndexedCorpus* corpus = NULL)
        : model_type(0), model_version(2), totaltokens(0), totaltypes(0), maxn(0), minn(999), hasskipgrams_(false), reverseindex(corpus), reverseindex_internal(false) {
        model_type = (unsigned char)this->getmodeltype();
    }
    /** load from stream / file (reference :670, :700) */
    PatternModel<ValueType, ValueHandler, MapType>(std::istream* f, const PatternModelOptions& options, PatternModelInterface* constrainmodel = NULL, IndexedCorpus* corpus = NULL)
        : PatternModel(corpus) {
        this->load(*f, options, constrainmodel);
    }
    PatternModel<ValueType, ValueHandler, MapType>(const std::string& filename, const PatternModelOptions& options, PatternModelInterface* constrainmodel = NULL,
                                                   IndexedCorpus* corpus = NULL)
        : PatternModel(corpus) {
        if (!options.QUIET) std::cerr << "Loading " << filename << std::endl;
        std::ifstream in(filename, std::ios::in | std::ios::binary);
        if (!in.good()) {
            std::cerr << "ERROR: Unable to load file " << filename << std::endl;
            throw InternalError();
        }
        this->load(in, options, constrainmodel);
    }
    virtual ~PatternModel() {
        if (reverseindex_internal) delete reverseindex;
    }

    int getmodeltype() const override { return colibri_host::is_indexed_value<ValueType>::value ? INDEXEDPATTERNMODEL : UNINDEXEDPATTERNMODEL; }
    int getmodelversion() const override { return 2; }
    PatternModelInterface* getinterface() { return (PatternModelInterface*)this; }

    size_t       size() const override { return MapType::size(); }
    bool         has(const Pattern& p) const override { return MapType::has(p); }
    bool         has(const PatternPointer& p) const override { return MapType::has(p); }
    int          maxlength() const override { return maxn; }
    int          minlength() const override { return minn; }
    unsigned int types() override { return (unsigned int)totaltypes; }
    unsigned int tokens() const override { return (unsigned int)totaltokens; }
    unsigned char type() const { return model_type; }
    unsigned char version() const { return model_version; }
    bool          hasskipgrams() const { return hasskipgrams_; }

    ValueType* getdata(const Pattern& pattern, bool makeifnew = false) {
        typename MapType::iterator it = this->find(pattern);
        if (it != this->end()) return &(it->second);
        if (makeifnew) return &((*this)[pattern]);
        return NULL;
    }
    unsigned int occurrencecount(const Pattern& pattern) override {
        ValueType* v = getdata(pattern, false);
        return v ? valuehandler.count(*v) : 0;
    }
    double frequency(const Pattern& pattern) override {  // occurrences over total tokens (coverage-free variant of reference :1697-1718)
        return totaltokens ? (double)occurrencecount(pattern) / (double)totaltokens : 0.0;
    }
    /** host-side add of one occurrence (reference :2059-2073); training itself never calls this, it is here for callers that extend a model */
    virtual void add(const Pattern& pattern, const IndexReference& ref) { valuehandler.add(getdata(pattern, true), ref); }
    /** erase patterns under a threshold, optionally only of size _n (reference :2107-2128) */
    unsigned int prune(int threshold, int _n = 0) {
        unsigned int pruned = 0;
        for (typename MapType::iterator it = this->begin(); it != this->end();) {
            if ((_n == 0 || (int)it->first.n() == _n) && (threshold == -1 || valuehandler.count(it->second) < (unsigned int)threshold)) {
                it = this->data.erase(it);
                ++pruned;
            } else {
                ++it;
            }
        }
        return pruned;
    }

    /**
     * Train on class-encoded corpus data — same signature as reference include/patternmodel.h:880-881.
     * `in` may be NULL when a preloaded corpus (reverse index) is attached.
     */
    virtual void train(std::istream* in, const PatternModelOptions& in_options, PatternModelInterface* constrainbymodel = NULL, PatternSet<>* filter = NULL, bool continued = false,
                       uint32_t firstsentence = 1, bool ignoreerrors = false) {
        (void)ignoreerrors;
        PatternModelOptions options = in_options;
        if (options.MINTOKENS == -1) options.MINTOKENS = 2;
        if (options.MINTOKENS == 0) options.MINTOKENS = 1;
        if (options.MINTOKENS_SKIPGRAMS < options.MINTOKENS) options.MINTOKENS_SKIPGRAMS = options.MINTOKENS;
        if (filter != NULL && filter->size() == 0) filter = NULL;  // cython passes empty sets (reference :902-903)
        if (constrainbymodel != NULL || filter != NULL || continued) {
            std::cerr << "ERROR: training constrained by another model, with a filter, or continued on a preloaded model is not on the MI355X-accelerated path" << std::endl;
            throw InternalError();
        }
        if (!this->data.empty() || result) {
            std::cerr << "ERROR: train() on a non-empty model is not on the MI355X-accelerated path" << std::endl;
            throw InternalError();
        }
        if (!options.QUIET) std::cerr << "Training patternmodel, occurrence threshold: " << options.MINTOKENS << std::endl;

        colibri_options o{};
        o.mintokens              = options.MINTOKENS;
        o.maxlength              = options.MAXLENGTH;
        o.minlength              = options.MINLENGTH;
        o.maxbackofflength       = options.MAXBACKOFFLENGTH;
        o.mintokens_unigrams     = options.MINTOKENS_UNIGRAMS;
        o.mintokens_skipgrams    = options.MINTOKENS_SKIPGRAMS;
        o.minskiptypes           = options.MINSKIPTYPES;
        o.maxskips               = options.MAXSKIPS;
        o.doskipgrams            = options.DOSKIPGRAMS;
        o.doskipgrams_exhaustive = options.DOSKIPGRAMS_EXHAUSTIVE;
        o.dopatternperline       = options.DOPATTERNPERLINE;
        o.prunenonsubsumed       = options.PRUNENONSUBSUMED;
        o.prunesubsumed          = options.PRUNESUBSUMED;
        o.indexed                = colibri_host::is_indexed_value<ValueType>::value ? 1 : 0;

        std::shared_ptr<colibri_host::TrainResult> r = std::make_shared<colibri_host::TrainResult>();
        if (reverseindex != NULL && !reverseindex->empty()) {
            colibri_host::device_train(reverseindex->beginpointer(), reverseindex->bytesize(), o, firstsentence, *r);
        } else if (in != NULL) {
            const std::vector<unsigned char> payload = colibri_host::read_corpus_payload(*in);
            if (payload.empty()) {
                std::cerr << "ERROR: Attempting to read pattern from file, but file is empty?" << std::endl;  // reference src/pattern.cpp:520-523
                throw InternalError();
            }
            colibri_host::device_train(payload.data(), payload.size(), o, firstsentence, *r);
        } else {
            std::cerr << "ERROR: No input stream and no reverse index (preloaded corpus) to train on" << std::endl;
            throw InternalError();
        }
        if (!options.QUIET) colibri_host::print_training_log(r->stats, o, std::cerr);
        totaltokens   = r->stats.totaltokens;
        totaltypes    = r->stats.totaltypes;
        if (r->stats.maxn > maxn) maxn = r->stats.maxn;
        if (r->stats.npatterns && r->stats.minn < minn) minn = r->stats.minn;
        hasskipgrams_ = (options.DOSKIPGRAMS || options.DOSKIPGRAMS_EXHAUSTIVE);
        install_result(r);
    }

    /** same, from a file name (reference :1353-1364); `.bz2` corpora are not accepted by this build */
    virtual void train(const std::string& filename, const PatternModelOptions& options, PatternModelInterface* constrainbymodel = NULL, PatternSet<>* filter = NULL,
                       bool continued = false, uint32_t firstsentence = 1, bool ignoreerrors = false) {
        if (filename.size() > 3 && filename.substr(filename.size() - 3) == ".bz2") {
            std::cerr << "ERROR: bz2-compressed corpora are not supported by this build; decompress first" << std::endl;
            throw InternalError();
        }
        std::ifstream in(filename, std::ios::in | std::ios::binary);
        if (!in.good() && !(reverseindex != NULL && !reverseindex->empty())) {
            std::cerr << "ERROR: Supplied data file can not be opened. Check whether it exists and whether you have proper permissions..." << std::endl;
            throw InternalError();
        }
        this->train(&in, options, constrainbymodel, filter, continued, firstsentence, ignoreerrors);
    }

    /** .colibri.patternmodel reader (reference :781-861 + PatternMapStore::read include/patternstore.h:555-619) */
    virtual void load(std::istream& f, const PatternModelOptions& options, PatternModelInterface* constrainmodel = NULL) {
        char null = 1;
        f.read(&null, 1);
        f.read((char*)&model_type, 1);
        f.read((char*)&model_version, 1);
        if (null != 0 || (model_type != UNINDEXEDPATTERNMODEL && model_type != INDEXEDPATTERNMODEL)) {
            std::cerr << "File is not a colibri model file (or a very old one, or a pointer/alignment model which this build does not read)" << std::endl;
            throw InternalError();
        }
        const unsigned char classencodingversion = model_version == 1 ? 1 : 2;
        f.read((char*)&totaltokens, sizeof(uint64_t));
        f.read((char*)&totaltypes, sizeof(uint64_t));
        uint64_t s = 0;
        f.read((char*)&s, sizeof(uint64_t));
        int mintokens = options.MINTOKENS == -1 ? 0 : options.MINTOKENS;
        this->data.clear();
        result.reset();
        this->pending_fill = nullptr;
        this->pending_size = nullptr;
        const bool file_indexed = model_type == INDEXEDPATTERNMODEL;
        for (uint64_t i = 0; i < s; ++i) {
            Pattern p = colibri_host::read_model_pattern(f, classencodingversion);
            ValueType value{};
            unsigned int cnt = 0;
            if (file_indexed) {
                IndexedData d;
                IndexedDataHandler().read(f, d);
                cnt = d.count();
                assign_loaded(value, d, cnt);
            } else {
                uint32_t c = 0;
                f.read((char*)&c, sizeof(uint32_t));
                cnt = c;
                IndexedData none;
                assign_loaded(value, none, c);
            }
            if (!f.good() && !f.eof()) {
                std::cerr << "ERROR: Exception occurred at pattern " << (i + 1) << " of " << s << std::endl;
                throw InternalError();
            }
            const PatternCategory c = p.category();
            if ((options.DOREMOVENGRAMS && c == NGRAM) || (options.DOREMOVESKIPGRAMS && c == SKIPGRAM) || (options.DOREMOVEFLEXGRAMS && c == FLEXGRAM)) continue;
            const int n = (int)p.n();
            if (n < options.MINLENGTH || n > options.MAXLENGTH) continue;
            if (cnt < (unsigned int)mintokens) continue;
            if (constrainmodel != NULL && !constrainmodel->has(p)) continue;
            if (options.DORESET) value = ValueType{};
            this->data[p] = value;
        }
        model_type    = (unsigned char)this->getmodeltype();
        model_version = 2;
        postread(options);
    }
    void load(const std::string& filename, const PatternModelOptions& options, PatternModelInterface* constrainmodel = NULL) {
        std::ifstream in(filename, std::ios::in | std::ios::binary);
        if (!in.good()) {
            std::cerr << "ERROR: Unable to load file " << filename << std::endl;
            throw InternalError();
        }
        this->load(in, options, constrainmodel);
    }
    /** recompute maxn / minn / hasskipgrams after loading (reference :572-588) */
    void postread(const PatternModelOptions&) {
        maxn = 0;
        minn = 999;
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            const int n = (int)it->first.n();
            if (n > maxn) maxn = n;
            if (n < minn) minn = n;
            if (!hasskipgrams_ && it->first.category() == SKIPGRAM) hasskipgrams_ = true;
        }
    }

    /** .colibri.patternmodel writer (reference :1609-1632, include/patternstore.h:534-542): 00, type, version 2, u64 tokens,
     *  u64 types, u64 npatterns, then per pattern: key bytes, 00, value. Streams straight from the device export when the map
     *  has not been materialised. */
    void write(std::ostream& out) {
        const char          null = 0;
        const unsigned char t = (unsigned char)this->getmodeltype(), v = 2;
        out.write(&null, 1);
        out.write((const char*)&t, 1);
        out.write((const char*)&v, 1);
        out.write((const char*)&totaltokens, sizeof(uint64_t));
        const uint64_t tp = this->types();
        out.write((const char*)&tp, sizeof(uint64_t));
        const uint64_t s = (uint64_t)this->size();
        out.write((const char*)&s, sizeof(uint64_t));
        if (result) {
            const colibri_host::TrainResult& r = *result;
            for (size_t j = 0; j < r.size(); ++j) {
                out.write((const char*)r.key_bytes.data() + r.key_off[j], (std::streamsize)(r.key_off[j + 1] - r.key_off[j]));
                out.write(&null, 1);
                colibri_host::write_value_from_result(out, r, j, (const ValueType*)NULL);
            }
        } else {
            for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
                it->first.write(out);
                valuehandler.write(out, it->second);
            }
        }
    }
    void write(const std::string& filename) {
        std::ofstream out(filename, std::ios::out | std::ios::binary);
        this->write(out);
    }

    /** one line per pattern: decoded text (or hex when no decoder), TAB, count — the core columns of reference print() (:2294-2340) */
    void print(std::ostream* out, const ClassDecoder* decoder = NULL) {
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            *out << (decoder ? it->first.tostring(*decoder) : it->first.tohex()) << "\t" << valuehandler.count(it->second);
            print_value_extra(*out, it->second);
            *out << std::endl;
        }
    }
    /** pattern and occurrence totals per order (a compact form of reference report(), :2500-2601) */
    void report(std::ostream* out) {
        std::vector<uint64_t> types_n(maxn + 2, 0), occ_n(maxn + 2, 0);
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            const size_t n = it->first.n();
            if (n < types_n.size()) {
                types_n[n] += 1;
                occ_n[n] += valuehandler.count(it->second);
            }
        }
        *out << "REPORT" << std::endl << "   Total word tokens in corpus: " << totaltokens << std::endl << "   Total word types in corpus:  " << totaltypes << std::endl;
        *out << "   Patterns in model: " << this->size() << std::endl << "   n\tpatterns\toccurrences" << std::endl;
        for (size_t n = 1; n < types_n.size(); ++n)
            if (types_n[n]) *out << "   " << n << "\t" << types_n[n] << "\t" << occ_n[n] << std::endl;
    }
    /** occurrence-count histogram (reference histogram(), :2603-2640) */
    void histogram(std::ostream* out) {
        std::map<unsigned int, uint64_t> hist;
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) hist[valuehandler.count(it->second)] += 1;
        *out << "HISTOGRAM" << std::endl << "occurrences\tpatterns" << std::endl;
        for (const auto& kv : hist) *out << kv.first << "\t" << kv.second << std::endl;
    }

  private:
    static void assign_loaded(uint32_t& dst, const IndexedData&, unsigned int count) { dst = count; }
    static void assign_loaded(IndexedData& dst, const IndexedData& src, unsigned int) { dst = src; }  // unindexed file -> indexed model: patterns load, counts are lost (reference :837-841)
    static void print_value_extra(std::ostream&, const uint32_t&) {}
    static void print_value_extra(std::ostream& out, const IndexedData& d) {
        out << "\t";
        bool first = true;
        for (const IndexReference& r : d.data) {
            if (!first) out << ' ';
            out << r.tostring();
            first = false;
        }
    }
};

/** Indexed model: pattern -> sorted list of (sentence, token). reference include/patternmodel.h:2682-3875. */
template <class MapType = PatternMap<IndexedData>>
class IndexedPatternModel : public PatternModel<IndexedData, IndexedDataHandler, MapType> {
  public:
    IndexedPatternModel<MapType>(IndexedCorpus* corpus = NULL) : PatternModel<IndexedData, IndexedDataHandler, MapType>(corpus) {}
    IndexedPatternModel<MapType>(const std::string& filename, const PatternModelOptions& options, PatternModelInterface* constrainmodel = NULL, IndexedCorpus* corpus = NULL)
        : PatternModel<IndexedData, IndexedDataHandler, MapType>(filename, options, constrainmodel, corpus) {}
    int getmodeltype() const override { return INDEXEDPATTERNMODEL; }

    void train(std::istream* in, const PatternModelOptions& options, PatternModelInterface* constrainbymodel = NULL, PatternSet<>* filter = NULL, bool continued = false,
               uint32_t firstsentence = 1, bool ignoreerrors = false) override {
        if (options.DOSKIPGRAMS && this->reverseindex == NULL) {  // reference :2828-2833
            std::cerr << "ERROR: You must specify a reverse index if you want to train skipgrams (or train skipgrams exhaustively)" << std::endl;
            throw InternalError();
        }
        PatternModel<IndexedData, IndexedDataHandler, MapType>::train(in, options, constrainbymodel, filter, continued, firstsentence, ignoreerrors);
    }
    void train(const std::string& filename, const PatternModelOptions& options, PatternModelInterface* constrainbymodel = NULL, PatternSet<>* filter = NULL, bool continued = false,
               uint32_t firstsentence = 1, bool ignoreerrors = false) override {
        if (options.DOSKIPGRAMS && this->reverseindex == NULL) {
            std::cerr << "ERROR: You must specify a reverse index if you want to train skipgrams (or train skipgrams exhaustively)" << std::endl;
            throw InternalError();
        }
        PatternModel<IndexedData, IndexedDataHandler, MapType>::train(filename, options, constrainbymodel, filter, continued, firstsentence, ignoreerrors);
    }
};
#endif
