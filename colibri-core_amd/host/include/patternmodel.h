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
#include <cstdlib>
#include <algorithm>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_set>
#include <vector>
#include <cstring>
#include <thread>

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
    /** (this build) the keys of all patterns, for installing the model as a device-side constraint set; false = cannot enumerate */
    virtual bool collect_keys(std::vector<uint64_t>&, std::vector<unsigned char>&) { return false; }
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
    std::shared_ptr<void>      device;  ///< (optional) the device context that still holds this model in HBM, for follow-up passes that need no host round trip
    size_t                     size() const { return counts.size(); }
    TrainResult()                              = default;
    TrainResult(const TrainResult&)            = default;
    TrainResult(TrainResult&&)                 = default;
    TrainResult& operator=(const TrainResult&) = default;
    TrainResult& operator=(TrainResult&&)      = default;
    ~TrainResult();  ///< the arrays go back to the process' result pool (colibri_host.cpp): the next export writes into pages that are already mapped
};

/** MINLENGTH > 1 in an unconstrained run: the reference counts the shorter orders (the look-back needs them) and prunes them away afterwards
 *  (patternmodel.h:1221-1230, :1339-1343), so the model is the full one without its patterns of fewer than `minlength` tokens */
void drop_short_patterns(TrainResult& r, int minlength);
/** whole corpus file -> v2 payload (header stripped; v1 data converted, reference src/classencoder.cpp:602-647) */
std::vector<unsigned char> read_corpus_payload(std::istream& in);
/** the keys a run is constrained to (colibri_set_constraint), or NULL */
struct ConstraintKeys {
    std::vector<uint64_t>      off;
    std::vector<unsigned char> bytes;
};
/** upload + train + export through the C ABI; prints the library's message on stderr and throws InternalError on any status != 0 */
void device_train(const unsigned char* payload, uint64_t nbytes, const colibri_options& opt, uint32_t firstsentence, TrainResult& out, const ConstraintKeys* constraint = NULL,
                  bool keep_device = false, bool continuation = false /* the keys are the model a continued run starts from (colibri_set_continuation) */,
                  bool as_filter = false /* the keys are train()'s filter (colibri_set_filter) */);
/** the same across `world` GPUs of this node (src/sharded.cpp): the corpus cut into contiguous sentence ranges, one device context and host thread per rank,
 *  RCCL for the exchange of candidate counts; the result is the union of the ranks' exports. Not for constrained runs and pattern lists. */
void device_train_sharded(const unsigned char* payload, uint64_t nbytes, const colibri_options& opt, uint32_t firstsentence, TrainResult& out, int world);
/** what the library keeps between plain train() calls of a process — the idle device context of every GPU (its working buffers stay reserved in HBM) and the arrays of the
 * last released model — given back now (a long-lived caller that is done training, or one that shares the GPU with other work) */
void release_cached();
/** how many GPUs train() uses: set_gpus(n) (the CLI's --gpus), else the environment's COLIBRI_GPUS, else 1 */
void set_gpus(int n);
int  gpus();
/** flexgrams abstracted from the skipgrams of an indexed model given in export layout (colibri_flexgrams + colibri_flexgrams_fetch) */
void device_flexgrams(const std::vector<uint64_t>& key_off, const unsigned char* key_bytes, const std::vector<uint64_t>& ref_off, const uint32_t* ref_sentence,
                      const uint16_t* ref_token, TrainResult& out);
/** the same on the model a device_train(..., keep_device = true) left resident (colibri_flexgrams_resident) */
void device_flexgrams_resident(const std::shared_ptr<void>& device, TrainResult& out);
/** the per-order progress lines the reference prints while training (patternmodel.h:1005-1019, :1195-1245) */
void print_training_log(const colibri_stats& s, const colibri_options& o, std::ostream& err);
/** the tokens of a key as byte strings, gaps included (what the reference's pattern.ngrams(…, 1) yields, src/pattern.cpp:1284-1296) */
void token_slices(const unsigned char* data, size_t bytes, std::vector<std::string>& out);
/** the column legends the reference writes to stderr after print() (:2306-2320, :2918-2927) and report() (:2591-2600) */
void print_legend(std::ostream& err, bool indexed);
void report_legend(std::ostream& err, bool indexed);
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
    // the 6-byte (u32 sentence, u16 token) records of the file format (reference include/datatypes.h:60-63), packed in blocks: one stream write per block
    // instead of two per reference
    unsigned char  buf[6 * 1024];
    const uint64_t a = r.ref_off.empty() ? 0 : r.ref_off[j];
    for (uint32_t k = 0; k < c;) {
        const uint32_t n = std::min<uint32_t>(c - k, 1024u);
        for (uint32_t q = 0; q < n; ++q) {
            std::memcpy(buf + 6 * q, &r.ref_sentence[a + k + q], 4);
            std::memcpy(buf + 6 * q + 4, &r.ref_token[a + k + q], 2);
        }
        out.write((const char*)buf, (std::streamsize)(6 * n));
        k += n;
    }
}
/** Look-ups on a device result that is still flat arrays: an open-addressed table (slot -> pattern number + 1) over the keys, built by several host threads the first
 *  time a caller asks for a pattern. The reference's callers look patterns up right after train() (src/test.cpp:1214-1232, src/benchmarks.cpp:232-236); turning the
 *  9.4 M patterns of a 10^8-token model into unordered_map nodes first cost 2.6-3.1 s (one heap Pattern per node), this costs tens of milliseconds. The node map is
 *  still built when a caller iterates, inserts or asks for an iterator / a non-trivial value object. */
class FlatIndex {
  public:
    explicit FlatIndex(const TrainResult& r) : r_(r) { build(); }
    /** pattern number of the key, or (size_t)-1 */
    size_t find(const unsigned char* key, size_t n) const {
        if (r_.size() == 0) return (size_t)-1;
        for (uint64_t s = hash(key, n) & mask_;; s = (s + 1) & mask_) {
            const uint32_t e = table_.get()[s];
            if (e == 0) return (size_t)-1;
            const size_t j = e - 1;
            if ((size_t)(r_.key_off[j + 1] - r_.key_off[j]) == n && std::memcmp(r_.key_bytes.data() + r_.key_off[j], key, n) == 0) return j;
        }
    }

  private:
    static uint64_t hash(const unsigned char* p, size_t n) {  // (any well-mixed hash: the table is private to this process)
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
        while (n >= 8) {
            uint64_t w;
            std::memcpy(&w, p, 8);
            h = (h ^ w) * 0xff51afd7ed558ccdULL;
            h ^= h >> 32;
            p += 8;
            n -= 8;
        }
        uint64_t w = 0;
        std::memcpy(&w, p, n);
        h = (h ^ w) * 0xc4ceb9fe1a85ec53ULL;
        h ^= h >> 29;
        h *= 0xff51afd7ed558ccdULL;
        return h ^ (h >> 32);
    }
    void build() {
        const size_t n = r_.size();
        if (n == 0 || n >= 0xFFFFFFFFull) {
            if (n) throw InternalError();
            return;
        }
        uint64_t slots = 64;
        while (slots < n + n / 2) slots <<= 1;
        mask_ = slots - 1;
        table_.reset(static_cast<uint32_t*>(std::calloc(slots, sizeof(uint32_t))));  // (zero pages on first touch: the threads below fault them in)
        if (!table_) throw std::bad_alloc();
        unsigned nt = std::thread::hardware_concurrency();
        nt          = std::max(1u, std::min(nt ? nt : 4u, 32u));
        if (n < 200000) nt = 1;
        uint32_t* const tab = table_.get();
        auto            job = [this, tab, n, nt](unsigned t) {
            const size_t a = n * t / nt, b = n * (t + 1) / nt;
            for (size_t j = a; j < b; ++j) {
                const unsigned char* k = r_.key_bytes.data() + r_.key_off[j];
                const size_t         m = (size_t)(r_.key_off[j + 1] - r_.key_off[j]);
                for (uint64_t s = hash(k, m) & mask_;; s = (s + 1) & mask_) {
                    uint32_t expect = 0;
                    if (__atomic_compare_exchange_n(&tab[s], &expect, (uint32_t)(j + 1), false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                }
            }
        };
        if (nt == 1) {
            job(0);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(job, t);
            for (auto& x : th) x.join();  // (join = the table's writes are visible to the caller)
        }
    }
    struct Free {
        void operator()(uint32_t* p) const { std::free(p); }
    };
    const TrainResult&              r_;
    std::unique_ptr<uint32_t, Free> table_;
    uint64_t                        mask_ = 0;
};

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
    bool          types_settled_ = false;  // a threshold-1 run already asked for its word types (reference :1201-1207): its statistics cache is no longer
                                           // empty, so types() does not compute them again — a pattern list without one-token lines keeps 0 types
    ValueHandler  valuehandler;
    std::shared_ptr<colibri_host::TrainResult> result;  // device results not yet turned into map nodes
    mutable std::shared_ptr<colibri_host::FlatIndex> flatindex;  // ... and the look-up table over them (built by the first has() / occurrencecount())
    /** pattern number in the pending device result, (size_t)-1 if absent; only meaningful while `result` is set */
    size_t flat_find(const Pattern& p) const { return flat_find(p.data, p.bytesize()); }
    /** the same for key bytes that are not a Pattern of their own (a PatternPointer's): no copy into a heap Pattern per look-up. The table is built once, under a lock:
     * read-only look-ups on a fresh model from several threads are as safe as the reference's */
    size_t flat_find(const unsigned char* bytes, size_t n) const {
        std::shared_ptr<colibri_host::FlatIndex> f = std::atomic_load(&flatindex);
        if (!f) {
            std::lock_guard<std::mutex> l(flat_mutex());
            f = std::atomic_load(&flatindex);
            if (!f) {
                f = std::make_shared<colibri_host::FlatIndex>(*result);
                std::atomic_store(&flatindex, f);
            }
        }
        return f->find(bytes, n);
    }
    static std::mutex& flat_mutex() {
        static std::mutex m;
        return m;
    }
    static unsigned int flat_count(const colibri_host::TrainResult& r, size_t j) {
        return colibri_host::is_indexed_value<ValueType>::value ? (r.ref_off.empty() ? 0u : (unsigned int)(r.ref_off[j + 1] - r.ref_off[j])) : (unsigned int)r.counts[j];
    }

    void install_result(std::shared_ptr<colibri_host::TrainResult> r) {
        result = r;
        flatindex.reset();
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
            self->flatindex.reset();
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
    bool         has(const Pattern& p) const override { return result ? flat_find(p) != (size_t)-1 : MapType::has(p); }
    bool         has(const PatternPointer& p) const override {
        if (result && p.mask == 0) return flat_find(p.data, (size_t)p.bytes) != (size_t)-1;  // (an unmasked pointer's bytes ARE the key: no heap Pattern per look-up)
        return this->has(Pattern(p));
    }
    int          maxlength() const override { return maxn; }
    int          minlength() const override { return minn; }
    unsigned int types() override {  // a loaded model without a type count falls back to the word types its patterns hold (reference :1700-1704)
        if (totaltypes == 0 && this->size() != 0 && !types_settled_) totaltypes = this->totalwordtypesingroup(0, 0);
        return (unsigned int)totaltypes;
    }
    unsigned int tokens() const override { return (unsigned int)totaltokens; }
    unsigned char type() const { return model_type; }
    unsigned char version() const { return model_version; }
    bool          hasskipgrams() const { return hasskipgrams_; }
    /** does nothing for unindexed models (reference :2653-2655); IndexedPatternModel abstracts its skipgrams on the device */
    virtual int computeflexgrams_fromskipgrams() { return 0; }
    /** does nothing for unindexed models (reference :2628) */
    virtual void outputrelations(const Pattern&, const ClassDecoder&, std::ostream&, const std::string& = "", bool = true) {}
    /** what the reference's constrained in-place rebuild leaves in the type count: the number of patterns the model was loaded with
     *  (it takes "total word types prior to pruning" from a map that already holds every pattern, patternmodel.h:1197-1201) */
    void settypes_inplace_rebuild() { totaltypes = this->size(); }

    bool collect_keys(std::vector<uint64_t>& off, std::vector<unsigned char>& bytes) override {
        off.assign(1, 0);
        bytes.clear();
        if (result) {  // device results not materialised yet: their flat arrays are exactly what is wanted
            off   = result->key_off;
            bytes = result->key_bytes;
            return true;
        }
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            const size_t n = it->first.bytesize();
            bytes.insert(bytes.end(), it->first.data, it->first.data + n);
            off.push_back(bytes.size());
        }
        return true;
    }
    ValueType* getdata(const Pattern& pattern, bool makeifnew = false) {
        typename MapType::iterator it = this->find(pattern);
        if (it != this->end()) return &(it->second);
        if (makeifnew) return &((*this)[pattern]);
        return NULL;
    }
    unsigned int occurrencecount(const Pattern& pattern) override {
        if (result) {  // a model fresh from the device: answered from its flat arrays
            const size_t j = flat_find(pattern);
            return j == (size_t)-1 ? 0u : flat_count(*result, j);
        }
        ValueType* v = getdata(pattern, false);
        return v ? valuehandler.count(*v) : 0;
    }
    /** the pattern's share of the occurrences of its own (category, size) group (reference :2047-2050) */
    double frequency(const Pattern& pattern) override { return this->occurrencecount(pattern) / (double)totaloccurrencesingroup((int)pattern.category(), (int)pattern.n()); }
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
        if (filter != NULL && constrainbymodel != NULL) filter = NULL;  // the reference only consults the filter in unconstrained runs (:1106)
        if (continued && (constrainbymodel != NULL || (this->data.empty() && !result))) continued = false;  // nothing to continue from / a constrained run counts everything anyway (:985)
        if (continued) {
            train_continued(in, options, firstsentence);
            return;
        }
        // Constrained training (reference :1062-1072, :1088-1089): one pass over all lengths, a window counts iff the constraint model has it.
        // The keys of the constraint model become a device-side set (colibri_set_constraint). constrainbymodel == this is the in-place
        // rebuild of patternmodeller -I / -2: the model's own patterns are the constraint, then it is emptied and rebuilt.
        const bool                   inplace = constrainbymodel == (PatternModelInterface*)this;
        colibri_host::ConstraintKeys ck;
        uint64_t                     constraint_tokens = 0, constraint_types = 0, loaded_patterns = 0;
        if (constrainbymodel != NULL) {
            if (options.DOSKIPGRAMS && !options.DOSKIPGRAMS_EXHAUSTIVE && !inplace) {  // reference :941-956
                options.DOSKIPGRAMS            = false;
                options.DOSKIPGRAMS_EXHAUSTIVE = true;
                if (!options.QUIET)
                    std::cerr << "WARNING: Skipgrams will be extracted exhaustively on the basis of the ngrams found; the constraint model will be applied only afterwards. "
                                 "This implies some skipgrams in the constraint model that are present may be missed, and it will not be most efficient. Use in-place "
                                 "rebuilding of your constraint model instead."
                              << std::endl;
            }
            if (inplace && (options.DOSKIPGRAMS || options.DOSKIPGRAMS_EXHAUSTIVE)) {
                std::cerr << "ERROR: skipgrams in an in-place constrained rebuild (trainskipgrams_selfconstrained) are not on the MI355X-accelerated path" << std::endl;
                throw InternalError();
            }
            if (!constrainbymodel->collect_keys(ck.off, ck.bytes)) {
                std::cerr << "ERROR: the constraint model cannot enumerate its patterns; pass a PatternModel / PatternSetModel of this build" << std::endl;
                throw InternalError();
            }
            if (inplace) {
                loaded_patterns = this->size();
                this->data.clear();
                result.reset();
                flatindex.reset();
                this->pending_fill = nullptr;
                this->pending_size = nullptr;
                maxn = 0;
                minn = 999;
            } else {
                constraint_tokens = constrainbymodel->tokens();  // the reference starts its totals from the constraint model's (:892-895)
                constraint_types  = constrainbymodel->types();
            }
        }
        if (!this->data.empty() || result) {
            std::cerr << "ERROR: train() on a non-empty model is not on the MI355X-accelerated path" << std::endl;
            throw InternalError();
        }
        colibri_host::ConstraintKeys fk;  // train(..., filter): the filter's patterns (reference :899-914, :1106-1133)
        if (filter != NULL) {
            if (options.DOSKIPGRAMS || options.DOSKIPGRAMS_EXHAUSTIVE || options.MINLENGTH > 1 || options.DOPATTERNPERLINE || options.MAXBACKOFFLENGTH < options.MAXLENGTH ||
                options.MINTOKENS_UNIGRAMS > options.MINTOKENS) {
                std::cerr << "ERROR: training with a filter is on the MI355X-accelerated path for MINLENGTH = 1, without skipgrams, back-off length, word threshold or pattern list"
                          << std::endl;
                throw InternalError();
            }
            bool hasngrams = false, hasother = false;
            fk.off.push_back(0);
            for (typename PatternSet<>::iterator it = filter->begin(); it != filter->end(); ++it) {
                (it->category() == NGRAM ? hasngrams : hasother) = true;
                fk.bytes.insert(fk.bytes.end(), it->data, it->data + it->bytesize());
                fk.off.push_back(fk.bytes.size());
            }
            if (!options.QUIET && hasngrams)
                std::cerr << "Filter with ngrams provided, only patterns that either match a filtered pattern or contain a smaller filtered pattern will be included..." << std::endl;
            if (!options.QUIET && hasother) std::cerr << "Filter with skipgrams provided, only matching instances will be included..." << std::endl;
        }
        if (!options.QUIET) std::cerr << "Training patternmodel, occurrence threshold: " << options.MINTOKENS << std::endl;

        colibri_options o{};
        o.mintokens              = options.MINTOKENS;
        o.maxlength              = options.MAXLENGTH;
        o.minlength              = (constrainbymodel == NULL) ? 1 : options.MINLENGTH;  // unconstrained: shorter patterns are dropped after the run (drop_short_patterns)
        o.maxbackofflength       = options.MAXBACKOFFLENGTH;
        o.mintokens_unigrams     = options.MINTOKENS_UNIGRAMS;
        o.mintokens_skipgrams    = options.MINTOKENS_SKIPGRAMS;
        o.minskiptypes           = options.MINSKIPTYPES;
        o.maxskips               = options.MAXSKIPS;
        o.doskipgrams            = options.DOSKIPGRAMS;
        o.doskipgrams_exhaustive = options.DOSKIPGRAMS_EXHAUSTIVE;
        o.dopatternperline       = options.DOPATTERNPERLINE;
        o.prunenonsubsumed       = 0;  // both subsumption prunes are post-hoc passes over the finished model (reference :1280-1330): done below, on the host
        o.prunesubsumed          = 0;
        o.indexed                = colibri_host::is_indexed_value<ValueType>::value ? 1 : 0;

        std::shared_ptr<colibri_host::TrainResult> r = std::make_shared<colibri_host::TrainResult>();
        // an indexed skipgram model stays resident on the device until it is materialised on the host: computeflexgrams_fromskipgrams works on it there
        const int  world = (constrainbymodel == NULL && filter == NULL && !options.DOPATTERNPERLINE) ? colibri_host::gpus() : 1;  // sentence-sharded over that many GPUs (src/sharded.cpp)
        const bool keep_device = world == 1 && o.indexed && o.doskipgrams && constrainbymodel == NULL && options.MINLENGTH <= 1;
        if (world > 1 || (world == 1 && constrainbymodel == NULL && filter == NULL && !options.DOPATTERNPERLINE && std::getenv("COLIBRI_GPUS_FORCE_SHARDED") != NULL)) {  // (forced: the sharded protocol on one rank, for tests; a filtered run has no sharded form)
            std::vector<unsigned char> owned;
            const unsigned char*       p = NULL;
            uint64_t                   nb = 0;
            if (reverseindex != NULL && !reverseindex->empty()) {
                p  = reverseindex->beginpointer();
                nb = reverseindex->bytesize();
            } else if (in != NULL) {
                owned = colibri_host::read_corpus_payload(*in);
                p     = owned.data();
                nb    = owned.size();
            }
            if (p == NULL || nb == 0) {
                std::cerr << "ERROR: No corpus data to train on" << std::endl;
                throw InternalError();
            }
            colibri_host::device_train_sharded(p, nb, o, firstsentence, *r, world);
        } else if (reverseindex != NULL && !reverseindex->empty()) {
            colibri_host::device_train(reverseindex->beginpointer(), reverseindex->bytesize(), o, firstsentence, *r, constrainbymodel ? &ck : (filter ? &fk : NULL), keep_device,
                                       false, filter != NULL);
        } else if (in != NULL) {
            const std::vector<unsigned char> payload = colibri_host::read_corpus_payload(*in);
            if (payload.empty()) {
                std::cerr << "ERROR: Attempting to read pattern from file, but file is empty?" << std::endl;  // reference src/pattern.cpp:520-523
                throw InternalError();
            }
            colibri_host::device_train(payload.data(), payload.size(), o, firstsentence, *r, constrainbymodel ? &ck : (filter ? &fk : NULL), keep_device, false, filter != NULL);
        } else {
            std::cerr << "ERROR: No input stream and no reverse index (preloaded corpus) to train on" << std::endl;
            throw InternalError();
        }
        if (constrainbymodel == NULL && options.MINLENGTH > 1) colibri_host::drop_short_patterns(*r, options.MINLENGTH);
        if (!options.QUIET) colibri_host::print_training_log(r->stats, o, std::cerr);
        totaltokens   = r->stats.totaltokens;
        totaltypes    = r->stats.totaltypes;
        if (constrainbymodel != NULL) {
            if (inplace) {
                // "total word types prior to pruning" is taken from a map that still holds every loaded pattern (:1197-1205): their number when
                // MINTOKENS > 1, the word types among the loaded unigrams when MINTOKENS == 1 (and MINLENGTH == 1), else left to types()
                if (options.MINTOKENS > 1) {
                    totaltypes = loaded_patterns;
                } else if (options.MINLENGTH == 1) {
                    uint64_t unigrams = 0;
                    for (size_t k = 0; k + 1 < ck.off.size(); ++k)
                        unigrams += colibri_host::token_count(ck.bytes.data() + ck.off[k], (size_t)(ck.off[k + 1] - ck.off[k])) == 1;
                    totaltypes = unigrams;
                } else {
                    totaltypes = 0;
                }
            } else {
                totaltokens += constraint_tokens;  // the corpus' tokens are added to the constraint model's total (:894, :1047-1048)
                totaltypes = constraint_types;
            }
        }
        if (r->stats.maxn > maxn) maxn = r->stats.maxn;
        if (r->stats.npatterns && r->stats.minn < minn) minn = r->stats.minn;
        hasskipgrams_ = (options.DOSKIPGRAMS || options.DOSKIPGRAMS_EXHAUSTIVE);
        types_settled_ = options.DOPATTERNPERLINE || (filter != NULL && options.MINTOKENS == 1);  // (a filter without one-token matches at threshold 1: 0 types, as above)
        install_result(r);
        if (options.PRUNENONSUBSUMED || options.PRUNESUBSUMED) prune_by_subsumption(options);
    }

  protected:
    /** train(..., continued = true) on a model that holds patterns (reference :983-995, colibri-patternmodeller -E): the orders the model already has n-grams
     *  of are skipped, the others are counted on the device with a look-back that finds the loaded patterns (colibri_set_continuation); the new patterns
     *  join the map, the totals stay the model's (:1047-1048 and :1197 are guarded by !continued). */
    void train_continued(std::istream* in, const PatternModelOptions& options, uint32_t firstsentence) {
        this->materialise();  // (a model trained a moment ago: its patterns into the map first)
        if (options.DOSKIPGRAMS || options.DOSKIPGRAMS_EXHAUSTIVE || options.MINTOKENS < 2 || options.MINLENGTH > 1 || options.DOPATTERNPERLINE ||
            options.MAXBACKOFFLENGTH < options.MAXLENGTH || options.MINTOKENS_UNIGRAMS > options.MINTOKENS) {
            std::cerr << "ERROR: continued training is on the MI355X-accelerated path for MINTOKENS >= 2, MINLENGTH = 1, without skipgrams, back-off length, word threshold or "
                         "pattern list" << std::endl;
            throw InternalError();
        }
        if (!options.QUIET) std::cerr << "Continuing training on preloaded model, computing statistics..." << std::endl;
        colibri_host::ConstraintKeys known;
        known.off.push_back(0);
        for (typename MapType::iterator it = this->data.begin(); it != this->data.end(); ++it) {
            if (it->first.category() != NGRAM) continue;
            known.bytes.insert(known.bytes.end(), it->first.data, it->first.data + it->first.bytesize());
            known.off.push_back(known.bytes.size());
        }
        colibri_options o{};
        o.mintokens           = options.MINTOKENS;
        o.maxlength           = options.MAXLENGTH;
        o.minlength           = 1;
        o.maxbackofflength    = options.MAXBACKOFFLENGTH;
        o.mintokens_unigrams  = options.MINTOKENS_UNIGRAMS;
        o.mintokens_skipgrams = options.MINTOKENS_SKIPGRAMS;
        o.minskiptypes        = options.MINSKIPTYPES;
        o.maxskips            = options.MAXSKIPS;
        o.indexed             = colibri_host::is_indexed_value<ValueType>::value ? 1 : 0;
        colibri_host::TrainResult r;
        if (reverseindex != NULL && !reverseindex->empty()) {
            colibri_host::device_train(reverseindex->beginpointer(), reverseindex->bytesize(), o, firstsentence, r, &known, false, /*continuation=*/true);
        } else if (in != NULL) {
            const std::vector<unsigned char> payload = colibri_host::read_corpus_payload(*in);
            if (payload.empty()) {
                std::cerr << "ERROR: Attempting to read pattern from file, but file is empty?" << std::endl;
                throw InternalError();
            }
            colibri_host::device_train(payload.data(), payload.size(), o, firstsentence, r, &known, false, /*continuation=*/true);
        } else {
            std::cerr << "ERROR: No input stream and no reverse index (preloaded corpus) to train on" << std::endl;
            throw InternalError();
        }
        if (!options.QUIET) colibri_host::print_training_log(r.stats, o, std::cerr);
        for (size_t j = 0; j < r.size(); ++j) {
            ValueType v{};
            colibri_host::value_from_result(r, j, v);
            this->data[Pattern(r.key_bytes.data() + r.key_off[j], (size_t)(r.key_off[j + 1] - r.key_off[j]))] = std::move(v);
        }
        if (r.stats.npatterns && r.stats.maxn > maxn) maxn = r.stats.maxn;
        for (int n = 1; n < COLIBRI_MAX_ORDER; ++n)
            if (r.stats.kept[n] && n < minn) minn = n;
    }

  public:
    /** erase the patterns of _n tokens (0 = any) that are not / are in the set (reference :2194-2242) */
    unsigned int prunenotinset(const std::unordered_set<Pattern>& s, int _n) { return prune_set(s, _n, false); }
    unsigned int pruneinset(const std::unordered_set<Pattern>& s, int _n) { return prune_set(s, _n, true); }

  protected:
    unsigned int prune_set(const std::unordered_set<Pattern>& s, int _n, bool erase_members) {
        unsigned int pruned = 0;
        if (s.empty()) return pruned;
        for (typename MapType::iterator it = this->begin(); it != this->end();) {
            if ((_n == 0 || (int)it->first.n() == _n) && ((s.find(it->first) != s.end()) == erase_members)) {
                it = this->data.erase(it);
                ++pruned;
            } else {
                ++it;
            }
        }
        return pruned;
    }
    /** the (n-1)-token sub-patterns of every pattern of n tokens */
    void subsumed_by(int n, std::unordered_set<Pattern>& out) {
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            if ((int)it->first.n() != n) continue;
            std::vector<std::string> toks;
            colibri_host::token_slices(it->first.data, it->first.bytesize(), toks);
            for (int first = 0; first + (n - 1) <= n; ++first) {
                std::string sub;
                for (int k = first; k < first + n - 1; ++k) sub += toks[(size_t)k];
                out.insert(Pattern((const unsigned char*)sub.data(), sub.size()));
            }
        }
    }
    /** PRUNENONSUBSUMED: from the longest patterns down, drop the (n-1)-grams no n-gram contains; PRUNESUBSUMED: from the shortest up, drop the
     *  (n-1)-grams some n-gram contains (reference :1280-1330). Post-hoc passes over the finished model, on the host. */
    void prune_by_subsumption(const PatternModelOptions& options) {
        if (options.PRUNENONSUBSUMED) {
            if (!options.QUIET) std::cerr << "Pruning non-subsumed n-grams" << std::endl;
            for (int n = std::min(options.PRUNENONSUBSUMED, options.MAXLENGTH); n > 1; n--) {
                std::unordered_set<Pattern> subsumed;
                subsumed_by(n, subsumed);
                const unsigned int k = prunenotinset(subsumed, n - 1);
                if (!options.QUIET) std::cerr << " pruned " << k << " non-subsumed " << (n - 1) << "-grams" << std::endl;
            }
        }
        if (options.PRUNESUBSUMED) {
            if (!options.QUIET) std::cerr << "Pruning subsumed n-grams" << std::endl;
            for (int n = 2; n <= std::min(options.PRUNESUBSUMED, options.MAXLENGTH); ++n) {
                std::unordered_set<Pattern> subsumed;
                subsumed_by(n, subsumed);
                const unsigned int k = pruneinset(subsumed, n - 1);
                if (!options.QUIET) std::cerr << " pruned " << k << " subsumed " << (n - 1) << "-grams" << std::endl;
            }
        }
    }

  public:

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
        flatindex.reset();
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

    // ---- views: print / report / histogram, text-identical to the reference's (goldens: tests/golden/views/) --------------------------------

    /** statistics per (category, size) group, 0 = all (reference computestats :1903-1935): occurrences and distinct patterns */
    void computestats() {
        cache_categories.clear();
        cache_n.clear();
        cache_grouptotal.clear();
        cache_grouptotalpatterns.clear();
        cache_categories.insert(0);
        cache_n.insert(0);
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            const int          c   = (int)it->first.category();
            const int          n   = (int)it->first.n();
            const unsigned int occ = valuehandler.count(it->second);
            cache_categories.insert(c);
            cache_n.insert(n);
            const int cs[2] = {c, 0}, ns[2] = {n, 0};
            for (int ci = 0; ci < 2; ++ci)
                for (int ni = 0; ni < 2; ++ni) {
                    if (ni == 0 && c == FLEXGRAM) continue;  // flexgrams have no per-size rows (:1918)
                    cache_grouptotal[cs[ci]][ns[ni]] += occ;
                    cache_grouptotalpatterns[cs[ci]][ns[ni]] += 1;
                }
        }
    }
    virtual void resetstats() {
        cache_grouptotalwordtypes.clear();
        cache_grouptotaltokens.clear();
        cache_coverage_done = false;
    }
    /**
     * word types and covered tokens per group (reference computecoveragestats :1946-1995). The reference walks the whole model once per group;
     * here every pattern is visited once and dropped into the (at most four) groups it belongs to. A group's word types are the distinct
     * tokens of its patterns — a skipgram's gap counts as one type, as it does there (pattern.ngrams(…,1) yields the gap token).
     * Unindexed models: the "covered tokens" of every group is the occurrence total of the whole model (the reference adds each pattern's count
     * outside its group filter, :1974), capped to tokens() by report().
     */
    virtual void computecoveragestats(int category = 0, int n = 0) {
        (void)category;
        (void)n;
        if (cache_coverage_done || this->size() == 0) return;
        if (cache_grouptotal.empty()) computestats();
        std::map<int, std::map<int, std::unordered_set<std::string>>> typesets;
        uint64_t alloccurrences = 0;
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            const int c = (int)it->first.category(), pn = (int)it->first.n();
            alloccurrences += valuehandler.count(it->second);
            std::vector<std::string> toks;
            colibri_host::token_slices(it->first.data, it->first.bytesize(), toks);
            const int cs[2] = {c, 0}, ns[2] = {pn, 0};
            for (int ci = 0; ci < 2; ++ci)
                for (int ni = 0; ni < 2; ++ni)
                    for (const std::string& t : toks) typesets[cs[ci]][ns[ni]].insert(t);
            this->coverage_visit(it, c, pn);
        }
        for (const int c : cache_categories)
            for (const int gn : cache_n) {
                cache_grouptotalwordtypes[c][gn] = (unsigned int)typesets[c][gn].size();
                cache_grouptotaltokens[c][gn]    = this->coverage_tokens(c, gn, alloccurrences);
            }
        this->coverage_finish();
        cache_coverage_done = true;
    }
    unsigned int totaloccurrencesingroup(int category, int n) {
        if (cache_grouptotal.empty() && this->size() != 0) computestats();
        return (unsigned int)cache_grouptotal[category][n];
    }
    unsigned int totalpatternsingroup(int category, int n) {
        if (cache_grouptotalpatterns.empty() && this->size() != 0) computestats();
        return (unsigned int)cache_grouptotalpatterns[category][n];
    }
    unsigned int totalwordtypesingroup(int category, int n) {
        if (cache_grouptotalwordtypes.empty() && this->size() != 0) computecoveragestats(category, n);
        return cache_grouptotalwordtypes[category][n];
    }
    unsigned int totaltokensingroup(int category, int n) {
        if (cache_grouptotaltokens.empty() && this->size() != 0) computecoveragestats(category, n);
        return (unsigned int)cache_grouptotaltokens[category][n];
    }
    /** occurrences × size: a maximal projection, also for indexed models (reference :1725-1727) */
    size_t coveragecount(const Pattern& key) { return (size_t)this->occurrencecount(key) * key.size(); }
    double coverage(const Pattern& key) { return coveragecount(key) / (double)this->tokens(); }

    /** the whole model, one pattern per line, under the reference's header (reference :2294-2321; the legend goes to stderr there and here) */
    virtual void print(std::ostream& out, const ClassDecoder& decoder, bool instantiate = false) {
        (void)instantiate;
        bool haveoutput = false;
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            if (!haveoutput) {
                out << "PATTERN\tCOUNT\tTOKENS\tCOVERAGE\tCATEGORY\tSIZE\tFREQUENCY" << (colibri_host::is_indexed_value<ValueType>::value ? "\tREFERENCES" : "") << std::endl;
                haveoutput = true;
            }
            this->print(out, decoder, it->first, instantiate, true);
        }
        if (haveoutput) colibri_host::print_legend(std::cerr, colibri_host::is_indexed_value<ValueType>::value);
    }
    /** one pattern: text, count, count×size, that over tokens(), category, size, frequency within its (category, size) group (reference :2354-2373, :2930-2959) */
    void print(std::ostream& out, const ClassDecoder& decoder, const Pattern& pattern, bool instantiate = false, bool endline = true) {
        (void)instantiate;
        const unsigned int count    = this->occurrencecount(pattern);
        const size_t       covcount = this->coveragecount(pattern);
        const int          cat      = (int)pattern.category();
        out << pattern.tostring(decoder) << "\t" << count << "\t" << covcount << "\t" << covcount / (double)this->tokens() << "\t"
            << (cat == NGRAM ? "ngram" : cat == SKIPGRAM ? "skipgram" : cat == FLEXGRAM ? "flexgram" : "") << "\t" << pattern.size() << "\t" << this->frequency(pattern);
        typename MapType::iterator it = this->find(pattern);
        if (it != this->end()) print_value_extra(out, it->second);
        if (endline) out << std::endl;
    }
    void printpattern(std::ostream& out, const ClassDecoder& decoder, const Pattern& pattern, bool instantiate = false, bool endline = true) {
        this->print(out, decoder, pattern, instantiate, endline);
    }
    /** pointer-style call kept for callers of the earlier form of this face: the same table; hex keys when there is no decoder */
    void print(std::ostream* out, const ClassDecoder* decoder = NULL) {
        if (decoder != NULL) {
            this->print(*out, *decoder, false);
            return;
        }
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            *out << it->first.tohex() << "\t" << valuehandler.count(it->second);
            print_value_extra(*out, it->second);
            *out << std::endl;
        }
    }

    /** occurrence count -> number of patterns (reference :2391-2410). cap keeps the top counts until `cap` patterns are covered. */
    void histogram(std::map<unsigned int, unsigned int>& hist, unsigned int threshold = 0, unsigned int cap = 0, int category = 0, unsigned int size = 0) {
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            if ((category != 0 && (int)it->first.category() != category) || (size != 0 && size != it->first.size())) continue;
            const unsigned int c = valuehandler.count(it->second);
            if (c >= threshold) hist[c]++;
        }
        if (cap > 0) {
            unsigned int sum = 0;
            std::map<unsigned int, unsigned int>::iterator cut = hist.end();
            while (sum < cap && cut != hist.begin()) {
                --cut;
                sum += cut->second;
            }
            hist.erase(hist.begin(), cut);
        }
    }
    void histogram(std::ostream& OUT, unsigned int threshold = 0, unsigned int cap = 0, int category = 0, unsigned int size = 0) {  // reference :2432-2441
        std::map<unsigned int, unsigned int> hist;
        histogram(hist, threshold, cap, category, size);
        OUT << "HISTOGRAM" << std::endl << "------------------------------" << std::endl << "OCCURRENCES\tPATTERNS" << std::endl;
        for (const auto& kv : hist) OUT << kv.first << "\t" << kv.second << std::endl;
    }
    void histogram(std::ostream* out) { histogram(*out); }
    /** smallest occurrence count among the top `amount` patterns (reference :2412-2422) */
    unsigned int topthreshold(int amount, int category = 0, int size = 0) {
        std::map<unsigned int, unsigned int> hist;
        histogram(hist, 0, (unsigned int)amount, category, (unsigned int)size);
        return hist.empty() ? 0 : hist.begin()->first;
    }

    /** the REPORT table (reference :2499-2601): totals, coverage, then one row per (category, size) group; leaves OUT in fixed/4 like the reference */
    void report(std::ostream& OUT, bool nocoverage = false) {
        const bool indexed = this->getmodeltype() != UNINDEXEDPATTERNMODEL;
        if (this->size() != 0) {
            if (nocoverage) {
                std::cerr << "Computing statistics without coverage information..." << std::endl;
                computestats();
            } else {
                std::cerr << "Computing statistics with coverage information (may take a while)..." << std::endl;
                computecoveragestats();
            }
        }
        const int W = 15;
        OUT << std::setiosflags(std::ios::fixed) << std::setprecision(4) << std::endl;
        OUT << "REPORT" << std::endl;
        if (!indexed && !nocoverage) {
            OUT << "   Warning: Model is unindexed, token coverage counts are mere maximal projections" << std::endl;
            OUT << "            assuming no overlap at all!!! Use an indexed model for accurate coverage counts" << std::endl;
        }
        OUT << "----------------------------------" << std::endl;
        OUT << "                          " << std::setw(W) << "PATTERNS" << std::setw(W) << "TOKENS" << std::setw(W) << "COVERAGE" << std::setw(W) << "TYPES" << std::setw(W) << std::endl;
        OUT << "Total:                    " << std::setw(W) << "-" << std::setw(W) << this->tokens() << std::setw(W) << "-" << std::setw(W) << this->types() << std::endl;
        if (!nocoverage) {
            const size_t coveredtypes  = totalwordtypesingroup(0, 0);
            size_t       coveredtokens = totaltokensingroup(0, 0);
            if (coveredtokens > this->tokens()) coveredtokens = this->tokens();
            const size_t uncoveredtokens = this->tokens() - coveredtokens;
            OUT << "Uncovered:                " << std::setw(W) << "-" << std::setw(W) << uncoveredtokens << std::setw(W) << uncoveredtokens / (double)this->tokens() << std::setw(W)
                << this->types() - coveredtypes << std::endl;
            OUT << "Covered:                  " << std::setw(W) << this->size() << std::setw(W) << coveredtokens << std::setw(W) << coveredtokens / (double)this->tokens() << std::setw(W)
                << coveredtypes << std::endl
                << std::endl;
        } else {
            OUT << std::endl;
        }
        bool haveoutput = false;
        for (const int c : cache_categories) {
            if (!cache_grouptotalpatterns.count(c)) continue;
            for (const int n : cache_n) {
                if (!cache_grouptotalpatterns[c].count(n)) continue;
                if (!haveoutput) {
                    OUT << std::setw(W) << "CATEGORY" << std::setw(W) << "N (SIZE) " << std::setw(W) << "PATTERNS";
                    if (indexed && !nocoverage) OUT << std::setw(W) << "TOKENS" << std::setw(W) << "COVERAGE";
                    if (!nocoverage) OUT << std::setw(W) << "TYPES";
                    OUT << std::setw(W) << "OCCURRENCES" << std::endl;
                    haveoutput = true;
                }
                OUT << std::setw(W) << (c == 0 ? "all" : c == NGRAM ? "n-gram" : c == SKIPGRAM ? "skipgram" : "flexgram");
                if (n == 0) OUT << std::setw(W) << "all";
                else OUT << std::setw(W) << n;
                OUT << std::setw(W) << cache_grouptotalpatterns[c][n];
                if (indexed && !nocoverage) OUT << std::setw(W) << cache_grouptotaltokens[c][n] << std::setw(W) << cache_grouptotaltokens[c][n] / (double)this->tokens();
                if (!nocoverage) OUT << std::setw(W) << cache_grouptotalwordtypes[c][n];
                OUT << std::setw(W) << cache_grouptotal[c][n] << std::endl;
            }
        }
        if (haveoutput) colibri_host::report_legend(std::cerr, indexed);
    }
    void report(std::ostream* out, bool nocoverage = false) { report(*out, nocoverage); }

  protected:
    std::set<int>                                  cache_categories, cache_n;
    std::map<int, std::map<int, uint64_t>>         cache_grouptotal, cache_grouptotalpatterns, cache_grouptotaltokens;
    std::map<int, std::map<int, unsigned int>>     cache_grouptotalwordtypes;
    bool                                           cache_coverage_done = false;
    // hooks the indexed model fills in to count really covered positions (reference :3390-3450)
    virtual void     coverage_visit(typename MapType::iterator, int, int) {}
    virtual uint64_t coverage_tokens(int, int, uint64_t alloccurrences) { return alloccurrences; }
    virtual void     coverage_finish() {}

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

/** The patterns of a model file as a set — what the reference loads a constraint model as (include/patternmodel.h:283-530; colibri-patternmodeller -j,
 *  src/patternmodeller.cpp:712-718). Here it is an unindexed model whose values are not looked at. */
class PatternSetModel : public PatternModel<uint32_t> {
  public:
    PatternSetModel() : PatternModel<uint32_t>() {}
    PatternSetModel(const std::string& filename, const PatternModelOptions& options, PatternModelInterface* constrainmodel = NULL) : PatternModel<uint32_t>(filename, options, constrainmodel) {}
    int getmodeltype() const override { return PATTERNSETMODEL; }
};

/** pattern -> relation count (reference include/patternmodel.h:220) */
typedef PatternMap<uint32_t> t_relationmap;

/** Indexed model: pattern -> sorted list of (sentence, token). reference include/patternmodel.h:2682-3875. */
template <class MapType = PatternMap<IndexedData>>
class IndexedPatternModel : public PatternModel<IndexedData, IndexedDataHandler, MapType> {
  public:
    IndexedPatternModel<MapType>(IndexedCorpus* corpus = NULL) : PatternModel<IndexedData, IndexedDataHandler, MapType>(corpus) {}
    IndexedPatternModel<MapType>(const std::string& filename, const PatternModelOptions& options, PatternModelInterface* constrainmodel = NULL, IndexedCorpus* corpus = NULL)
        : PatternModel<IndexedData, IndexedDataHandler, MapType>(filename, options, constrainmodel, corpus) {}
    int getmodeltype() const override { return INDEXEDPATTERNMODEL; }

  protected:
    // Really covered positions (reference :3390-3450): every occurrence covers size() consecutive positions, gaps included. The reference
    // fills this only for the "all sizes" rows — its per-size test compares the *argument* n (0 from report()) with the group's size
    // (:3425) — so per-size rows report 0 tokens there, and here.
    std::map<int, std::vector<uint64_t>> covered_;
    void coverage_visit(typename PatternMap<IndexedData>::iterator it, int c, int pn) override {
        std::vector<uint64_t>&bycat = covered_[c], &all = covered_[0];
        for (const IndexReference& r : it->second.data)
            for (int i = 0; i < pn; ++i) {
                const uint64_t pos = ((uint64_t)r.sentence << 16) | (uint16_t)(r.token + i);
                bycat.push_back(pos);
                all.push_back(pos);
            }
    }
    uint64_t coverage_tokens(int c, int gn, uint64_t) override {
        if (gn != 0) return 0;
        std::vector<uint64_t>& v = covered_[c];
        std::sort(v.begin(), v.end());
        return (uint64_t)(std::unique(v.begin(), v.end()) - v.begin());
    }
    void coverage_finish() override { covered_.clear(); }

    // (length, gap mask) of every skipgram in the model, by length: what the reference's matchskipgramhelper (:1722-1744) is used for — a
    // skipgram of the model can only match a window if its first word is the window's, so testing every mask of that length with has() finds the same set
    std::map<int, std::vector<uint32_t>> skipmasks_;
    size_t                                skipmasks_size_ = (size_t)-1;
    void compute_skipmasks() {
        if (skipmasks_size_ == this->size()) return;
        skipmasks_.clear();
        std::set<std::pair<int, uint32_t>> seen;
        for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
            if (it->first.category() != SKIPGRAM) continue;
            std::vector<std::string> toks;
            colibri_host::token_slices(it->first.data, it->first.bytesize(), toks);
            uint32_t mask = 0;
            for (size_t k = 0; k < toks.size() && k < 31; ++k)
                if (toks[k].size() == 1 && (unsigned char)toks[k][0] == colibri_classes::skipclass) mask |= (1u << k);
            if (seen.insert(std::make_pair((int)toks.size(), mask)).second) skipmasks_[(int)toks.size()].push_back(mask);
        }
        skipmasks_size_ = this->size();
    }
    void need_reverseindex() const {
        if (this->reverseindex == NULL || this->reverseindex->empty()) {
            std::cerr << "ERROR: No reverse index present" << std::endl;
            throw InternalError();
        }
    }
    /** PatternPointer::operator==(const Pattern&) of the reference for a skipgram pointer (src/pattern.cpp:1009-1041), byte for byte */
    static bool masked_pointer_equals(const unsigned char* data, size_t bytes, uint32_t mask, const Pattern& other) {
        const size_t obytes = other.bytesize();
        auto         at     = [&](size_t i) -> unsigned char { return i < obytes ? other.data[i] : 0; };
        if (bytes == 0 || data[0] == 0) return obytes == 0;
        if (obytes == 0) return false;
        size_t tok = 0;
        for (size_t i = 0; i < bytes; ++i) {
            if (i > 0 && at(i - 1) >= 128 && at(i) == 0) return false;
            if (mask != 0 && data[i] < 128) {
                if (tok <= 30 && (mask & (1u << tok))) {
                    if (at(i) != colibri_classes::skipclass) return false;
                } else if (data[i] != at(i)) {
                    return false;
                }
                ++tok;
            } else if (data[i] != at(i)) {
                return false;
            }
        }
        return at(bytes) == colibri_classes::delimiterclass;
    }
    static void prunerelations(t_relationmap& relations, unsigned int occurrencethreshold) {  // reference :3066-3078
        for (t_relationmap::iterator it = relations.begin(); it != relations.end();) {
            if (it->second < occurrencethreshold) it = relations.erase(it);
            else ++it;
        }
    }

  public:
    /** The skip content of a skipgram (or flexgram): for every occurrence, the tokens of the corpus from the pattern's first gap to its last
     *  gap, as a plain n-gram — the slice drops the mask (reference :3029-3059; src/pattern.cpp:853-855). Plain host code over the forward
     *  index and the loaded corpus, as in the reference: a per-pattern query, not a corpus pass. */
    t_relationmap getskipcontent(const Pattern& pattern) {
        t_relationmap skipcontent;
        if (this->reverseindex == NULL) {
            std::cerr << "ERROR: No corpus data loaded! (in PatternModel::getskipcontent)" << std::endl;
            throw InternalError();
        }
        if (pattern.category() == NGRAM) return skipcontent;
        IndexedData* data = this->getdata(pattern);
        if (data == NULL) throw NoSuchPattern();
        std::vector<std::string> toks;
        colibri_host::token_slices(pattern.data, pattern.bytesize(), toks);
        const int n = (int)toks.size();
        auto      isgap = [&](int k) { return toks[(size_t)k].size() == 1 && ((unsigned char)toks[(size_t)k][0] == colibri_classes::skipclass || (unsigned char)toks[(size_t)k][0] == colibri_classes::flexclass); };
        int       head = 0, tail = 0;  // leading / trailing tokens that are no gaps (maskheadskip / masktailskip of the reversed mask, src/algorithms.cpp:56-77)
        while (head < n && !isgap(head)) ++head;
        while (tail < n - head && !isgap(n - tail - 1)) ++tail;
        for (const IndexReference& ref : data->data) {
            const PatternPointer     raw = this->reverseindex->getpattern(ref, n);
            std::vector<std::string> w;
            colibri_host::token_slices(raw.data, raw.bytesize(), w);
            std::string content;
            for (int k = head; k < n - tail; ++k) content += w[(size_t)k];
            skipcontent[Pattern((const unsigned char*)content.data(), content.size())] += 1;
        }
        return skipcontent;
    }
    /** the n-grams of the model that instantiate the given skipgram / flexgram at its occurrences (reference :3127-3165 over getreverseindex
     *  :1746-1824 with category NGRAM: the window of the pattern's own length at each occurrence, if the model has it) */
    t_relationmap getinstances(const Pattern& pattern, unsigned int occurrencethreshold = 0) {
        need_reverseindex();
        IndexedData* data = this->getdata(pattern);
        if (data == NULL) throw NoSuchPattern();
        t_relationmap instances;
        const int     n = (int)pattern.n();
        if (n >= this->minlength() && n <= this->maxlength()) {
            for (const IndexReference& ref : data->data) {
                if (ref.token + (unsigned int)n > this->reverseindex->sentencelength((int)ref.sentence)) continue;
                const Pattern candidate(this->reverseindex->getpattern(ref, n));
                if (candidate.category() != NGRAM || candidate == pattern) continue;
                if (occurrencethreshold == 0 ? !this->has(candidate) : this->occurrencecount(candidate) < occurrencethreshold) continue;
                instances[candidate] += 1;
            }
        }
        if (occurrencethreshold > 0) prunerelations(instances, occurrencethreshold);
        return instances;
    }
    /** the skipgrams of the model that abstract over the given pattern at its occurrences (reference :3086-3118): every gap mask a skipgram
     *  of this length has in the model, laid over the window at each occurrence */
    t_relationmap gettemplates(const Pattern& pattern, unsigned int occurrencethreshold = 0) {
        need_reverseindex();
        IndexedData* data = this->getdata(pattern);
        if (data == NULL) throw NoSuchPattern();
        t_relationmap templates;
        const int     n = (int)pattern.n();
        if (this->hasskipgrams() && n >= 3 && n >= this->minlength() && n <= this->maxlength() &&
            (occurrencethreshold == 0 || this->occurrencecount(pattern) >= occurrencethreshold)) {
            compute_skipmasks();
            const std::vector<uint32_t>& masks = skipmasks_[n];
            for (const IndexReference& ref : data->data) {
                if (ref.token + (unsigned int)n > this->reverseindex->sentencelength((int)ref.sentence)) continue;
                PatternPointer window = this->reverseindex->getpattern(ref, n);
                for (uint32_t mask : masks) {
                    window.mask = mask;
                    const Pattern candidate(window);
                    if (candidate.category() != SKIPGRAM || !this->has(candidate)) continue;
                    // The reference drops the pattern itself with `candidate != pattern`: a masked pointer into the corpus against the
                    // materialised pattern, compared byte by byte AT THE SAME INDEX (src/pattern.cpp:1009-1041). That recognises the pattern only
                    // where every token under a gap is a single byte (elsewhere a skipgram is listed as its own template), and it takes an
                    // n-gram for the candidate when the low byte of the token under the gap happens to be 03 — reproduced, not corrected.
                    if (masked_pointer_equals(window.data, (size_t)window.bytesize(), mask, pattern)) continue;
                    templates[candidate] += 1;
                }
            }
        }
        if (occurrencethreshold > 0) prunerelations(templates, occurrencethreshold);
        return templates;
    }
    /** one row per related pattern (reference :3595-3609) */
    void outputrelations(const Pattern& pattern, t_relationmap& relations, const ClassDecoder& classdecoder, std::ostream& OUT, const std::string& label = "RELATED-TO") {
        int total = 0;
        for (t_relationmap::iterator it = relations.begin(); it != relations.end(); ++it) total += (int)it->second;
        if (total == 0) return;
        const double      total_f   = total;
        const std::string pattern_s = pattern.tostring(classdecoder);
        for (t_relationmap::iterator it = relations.begin(); it != relations.end(); ++it)
            OUT << "\t" << pattern_s << "\t" << label << "\t" << it->first.tostring(classdecoder) << "\t" << it->second << "\t" << it->second / total_f << "\t"
                << this->occurrencecount(it->first) << std::endl;
    }
    /** the relations of a pattern selected by `filter` (reference :3622-3662). Built here: skipcontent. "instances" and "templates" print
     *  nothing through this entry point in the reference either — its call with a PatternPointer resolves to the base class's empty
     *  getinstances / gettemplates(const PatternPointer&) (:2635-2640), the working ones take a Pattern (used directly, src/test.cpp:1457) */
    void outputrelations(const Pattern& pattern, const ClassDecoder& classdecoder, std::ostream& OUT, const std::string& filter = "", bool outputheader = true) override {
        if (filter != "skipcontent" && filter != "instances" && filter != "templates") {
            std::cerr << "ERROR: relation '" << (filter.empty() ? "all" : filter) << "' is not part of the MI355X-accelerated build (see DESIGN.md, out of scope)" << std::endl;
            throw InternalError();
        }
        if (outputheader) OUT << "#\tPATTERN1\tRELATION\tPATTERN2\tREL.COUNT\tREL.FREQUENCY\tCOUNT2" << std::endl;
        if (filter == "skipcontent") {
            t_relationmap relations = this->getskipcontent(pattern);
            this->outputrelations(pattern, relations, classdecoder, OUT, "INSTANTIATED-BY");
        }
    }
    /** Compute flexgrams by abstracting from the skipgrams in the model (reference :3724-3744): every skipgram's references are appended to
     *  the flexgram it abstracts to (Pattern::toflexgram). The group-by and the merge of the reference lists run on the device
     *  (colibri_flexgrams); each flexgram's new references arrive ascending. @return the number of flexgrams that were not in the model */
    int computeflexgrams_fromskipgrams() override {
        colibri_host::TrainResult flex;
        if (this->result && this->result->device) {  // the model is still in HBM: group and merge there, only the flexgrams come back
            colibri_host::device_flexgrams_resident(this->result->device, flex);
            this->result->device.reset();
        } else if (this->result) {  // device results not materialised yet: their flat arrays are the input
            std::shared_ptr<colibri_host::TrainResult> r = this->result;
            colibri_host::device_flexgrams(r->key_off, r->key_bytes.data(), r->ref_off, r->ref_sentence.data(), r->ref_token.data(), flex);
        } else {
            std::vector<uint64_t>      key_off(1, 0), ref_off(1, 0);
            std::vector<unsigned char> key_bytes;
            std::vector<uint32_t>      rs;
            std::vector<uint16_t>      rt;
            for (typename MapType::iterator it = this->begin(); it != this->end(); ++it) {
                if (it->first.category() != SKIPGRAM) continue;
                key_bytes.insert(key_bytes.end(), it->first.data, it->first.data + it->first.bytesize());
                key_off.push_back(key_bytes.size());
                for (const IndexReference& ref : it->second.data) {
                    rs.push_back(ref.sentence);
                    rt.push_back(ref.token);
                }
                ref_off.push_back(rs.size());
            }
            rs.push_back(0);
            rt.push_back(0);
            colibri_host::device_flexgrams(key_off, key_bytes.data(), ref_off, rs.data(), rt.data(), flex);
        }
        if (this->result) {
            // the model has not been turned into map nodes yet: the flexgrams are appended to its flat arrays (no pattern of a freshly trained model
            // is a flexgram — a corpus with a literal {**} token is refused at upload — so every one of them is new)
            colibri_host::TrainResult& r   = *this->result;
            this->flatindex.reset();  // (built over the arrays as they were)
            const size_t               np  = r.size(), nf = flex.size();
            const uint64_t             kb  = r.key_off[np], nr = r.ref_off[np], fkb = flex.key_off[nf], fnr = flex.ref_off[nf];
            r.key_bytes.resize((size_t)(kb + fkb) + 1);
            std::memcpy(r.key_bytes.data() + kb, flex.key_bytes.data(), (size_t)fkb);
            r.ref_sentence.resize((size_t)(nr + fnr) + 1);
            r.ref_token.resize((size_t)(nr + fnr) + 1);
            std::memcpy(r.ref_sentence.data() + nr, flex.ref_sentence.data(), (size_t)fnr * sizeof(uint32_t));
            std::memcpy(r.ref_token.data() + nr, flex.ref_token.data(), (size_t)fnr * sizeof(uint16_t));
            r.key_off.resize(np + nf + 1);
            r.ref_off.resize(np + nf + 1);
            r.counts.resize(np + nf);
            for (size_t j = 0; j < nf; ++j) {
                r.key_off[np + j + 1] = kb + flex.key_off[j + 1];
                r.ref_off[np + j + 1] = nr + flex.ref_off[j + 1];
                r.counts[np + j]      = flex.counts[j];
            }
            r.stats.npatterns = np + nf;
            return (int)nf;
        }
        int count = 0;
        for (size_t j = 0; j < flex.size(); ++j) {
            const Pattern flexgram(flex.key_bytes.data() + flex.key_off[j], (size_t)(flex.key_off[j + 1] - flex.key_off[j]));
            if (!this->has(flexgram)) ++count;
            IndexedData& d = (*this)[flexgram];
            d.data.reserve(d.data.size() + (size_t)(flex.ref_off[j + 1] - flex.ref_off[j]));
            for (uint64_t k = flex.ref_off[j]; k < flex.ref_off[j + 1]; ++k) d.data.push_back(IndexReference(flex.ref_sentence[k], flex.ref_token[k]));
        }
        return count;
    }
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
