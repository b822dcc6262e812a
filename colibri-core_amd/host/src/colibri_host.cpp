// colibri_host.cpp — non-template parts of the C++ face: key-type helpers, corpus / model file formats, and the glue
// that drives libcolibri_hip.so through its C ABI (include/colibri_hip.h). No counting happens here.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <map>
#include <mutex>
#include <sstream>

#include "algorithms.h"
#include "patternmodel.h"
#include "spooky_device.hpp"  // the same SpookyHash routine the kernels use, compiled for the host

// ---------------------------------------------------------------------------------------------------
// key helpers
// ---------------------------------------------------------------------------------------------------
namespace colibri_host {

uint64_t spooky_hash64(const unsigned char* data, size_t len) {
    if (len >= 192) {
        std::cerr << "ERROR: pattern of " << len << " bytes: SpookyHash's long form is not part of this build" << std::endl;
        throw InternalError();
    }
    unsigned char buf[192 + 16] = {0};  // the device routine may read 15 bytes past the key
    std::memcpy(buf, data, len);
    return colibri::spooky64_short(buf, (uint32_t)len);
}

size_t token_count(const unsigned char* data, size_t bytes) {
    size_t n = 0;
    for (size_t i = 0; i < bytes; ++i) n += data[i] < 128;
    return n;
}

size_t key_bytesize(const unsigned char* data) {
    size_t i        = 0;
    bool   prevhigh = false;
    while (prevhigh || data[i] != 0) {
        prevhigh = data[i] >= 128;
        ++i;
    }
    return i;
}

PatternCategory category_of(const unsigned char* data, size_t bytes) {
    bool prevhigh = false;
    for (size_t i = 0; i < bytes; ++i) {
        if (!prevhigh && data[i] == colibri_classes::flexclass) return FLEXGRAM;
        if (!prevhigh && data[i] == colibri_classes::skipclass) return SKIPGRAM;
        prevhigh = data[i] >= 128;
    }
    return NGRAM;
}

std::string decode_key(const unsigned char* data, size_t bytes, const ClassDecoder& decoder) {
    std::string out;
    size_t      i = 0;
    while (i < bytes) {
        unsigned int len = 0;
        const unsigned int cls = bytestoint(data + i, &len);
        if (!out.empty()) out += ' ';
        if (decoder.hasclass(cls)) {
            out += decoder[cls];
        } else {
            out += "{?}";
        }
        i += len;
    }
    return out;
}

}  // namespace colibri_host

Pattern::Pattern(const PatternPointer& pp) {
    if (pp.data == NULL || pp.bytes == 0) {
        data = NULL;
        return;
    }
    if (pp.mask == 0) {
        assign(pp.data, pp.bytes);
        return;
    }
    // skipgram / flexgram: every gapped token becomes one marker byte; adjacent flex gaps collapse (reference src/pattern.cpp:886-908, :1043-1065)
    const bool                 flex = pp.isflexgram();
    std::vector<unsigned char> buf;
    buf.reserve(pp.bytes);
    int  tok     = 0;
    bool prevgap = false;
    size_t start = 0;
    for (size_t i = 0; i < pp.bytes; ++i) {
        if (pp.data[i] < 128) {
            if (pp.isgap(tok)) {
                if (!(flex && prevgap)) buf.push_back(flex ? colibri_classes::flexclass : colibri_classes::skipclass);
                prevgap = true;
            } else {
                buf.insert(buf.end(), pp.data + start, pp.data + i + 1);
                prevgap = false;
            }
            ++tok;
            start = i + 1;
        }
    }
    assign(buf.data(), buf.size());
}

Pattern::Pattern(std::istream& in, bool ignoreeol, const unsigned char version) {
    (void)ignoreeol;
    data            = NULL;
    const Pattern p = colibri_host::read_model_pattern(in, version);
    assign(p.data, p.data ? p.bytesize() : 0);
}

void Pattern::write(std::ostream& out, const unsigned char*) const {
    const size_t s = bytesize();
    if (s > 0) {
        out.write((const char*)data, (std::streamsize)s + 1);
    } else {
        const char null = 0;
        out.write(&null, 1);
    }
}

std::string Pattern::tostring(const ClassDecoder& decoder) const { return data ? colibri_host::decode_key(data, bytesize(), decoder) : std::string(); }

std::string Pattern::tohex() const {
    static const char* d = "0123456789abcdef";
    std::string        s;
    const size_t       b = bytesize();
    for (size_t i = 0; i < b; ++i) {
        s.push_back(d[data[i] >> 4]);
        s.push_back(d[data[i] & 15]);
    }
    return s;
}

namespace colibri_host {
void token_slices(const unsigned char* data, size_t bytes, std::vector<std::string>& out) {
    size_t begin = 0;
    for (size_t i = 0; i < bytes; ++i)
        if (data[i] < 128) {  // a byte under 128 closes a token
            out.emplace_back((const char*)data + begin, i + 1 - begin);
            begin = i + 1;
        }
}
void print_legend(std::ostream& err, bool indexed) {
    err << std::endl << "Legend:" << std::endl;
    err << " - PATTERN    : The pattern, Gaps in skipgrams are represented as {*}. Variable-width gaps in flexgrams are shown using {**}." << std::endl;
    err << " - COUNT      : The occurrence count - the amount of times the pattern occurs in the data" << std::endl;
    if (indexed) {
        err << " - TOKENS     : The number of tokens in the corpus that this pattern covers" << std::endl;
        err << " - COVERAGE   : The number of tokens covered, as a fraction of the total in the corpus" << std::endl;
    } else {
        err << " - TOKENS     : The maximum number of tokens in the corpus that this pattern covers (a projection: count x size, the model is not indexed)" << std::endl;
        err << " - COVERAGE   : The maximum number of tokens covered, as a fraction of the total in the corpus (projection)" << std::endl;
    }
    err << " - CATEGORY   : The pattern type category (ngram,skipgram,flexgram)" << std::endl;
    err << " - SIZE       : The size of the pattern (in tokens)" << std::endl;
    err << " - FREQUENCY  : The frequency of the pattern within its pattern type category and size-class." << std::endl;
    err << " - REFERENCES : A space-delimited list of sentence:token position where the pattern occurs in the data. Sentences start at 1, tokens at 0" << std::endl;
}
void report_legend(std::ostream& err, bool indexed) {
    err << std::endl << "Legend:" << std::endl;
    err << " - PATTERNS    : The number of distinct patterns within the group" << std::endl;
    if (indexed) {
        err << " - TOKENS      : The number of tokens that is covered by the patterns in the group." << std::endl;
        err << " - COVERAGE    : The number of tokens covered, as a fraction of the total in the corpus" << std::endl;
    }
    err << " - TYPES       : The number of unique *word/unigram* types in this group" << std::endl;
    err << " - OCCURRENCES : The total number of occurrences of the patterns in this group" << std::endl;
}
}  // namespace colibri_host

int PatternPointer::ngrams(std::vector<std::pair<PatternPointer, int>>& container, const int n) const {
    std::vector<size_t> starts{0};
    for (size_t i = 0; i < bytes; ++i)
        if (data[i] < 128) starts.push_back(i + 1);
    const int ntok = (int)starts.size() - 1;
    if (n > ntok) return 0;
    for (int i = 0; i + n <= ntok; ++i) container.push_back(std::make_pair(PatternPointer(data + starts[i], starts[i + n] - starts[i]), i));
    return ntok - n + 1;
}
int PatternPointer::ngrams(std::vector<PatternPointer>& container, const int n) const {
    std::vector<std::pair<PatternPointer, int>> tmp;
    const int                                   r = ngrams(tmp, n);
    for (const auto& p : tmp) container.push_back(p.first);
    return r;
}
namespace {
// the sizes minn..maxn one after the other, each through the container's own ngrams()
template <class Source, class Container>
int sub_windows(const Source& src, Container& container, int minn, int maxn) {
    const int ntok = (int)src.n();
    if (maxn > ntok) maxn = ntok;
    if (minn > ntok) return 0;
    int found = 0;
    for (int k = minn; k <= maxn; ++k) found += src.ngrams(container, k);
    return found;
}
}  // namespace
int PatternPointer::subngrams(std::vector<PatternPointer>& container, int minn, int maxn) const { return sub_windows(*this, container, minn, maxn); }
int PatternPointer::subngrams(std::vector<std::pair<PatternPointer, int>>& container, int minn, int maxn) const { return sub_windows(*this, container, minn, maxn); }

// Pattern: the windows are views into (or copies of) this pattern's own bytes; a gap token (the byte 03) stays what it is
int Pattern::ngrams(std::vector<std::pair<PatternPointer, int>>& container, const int n) const {
    if (data == NULL) return 0;
    return PatternPointer(data, bytesize()).ngrams(container, n);
}
int Pattern::ngrams(std::vector<PatternPointer>& container, const int n) const {
    if (data == NULL) return 0;
    return PatternPointer(data, bytesize()).ngrams(container, n);
}
int Pattern::ngrams(std::vector<std::pair<Pattern, int>>& container, const int n) const {
    std::vector<std::pair<PatternPointer, int>> tmp;
    const int                                   r = ngrams(tmp, n);
    for (const auto& p : tmp) container.push_back(std::make_pair(Pattern(p.first.data, p.first.bytes), p.second));
    return r;
}
int Pattern::ngrams(std::vector<Pattern>& container, const int n) const {
    std::vector<std::pair<PatternPointer, int>> tmp;
    const int                                   r = ngrams(tmp, n);
    for (const auto& p : tmp) container.push_back(Pattern(p.first.data, p.first.bytes));
    return r;
}
int Pattern::subngrams(std::vector<Pattern>& container, int minn, int maxn) const { return sub_windows(*this, container, minn, maxn); }
int Pattern::subngrams(std::vector<PatternPointer>& container, int minn, int maxn) const { return sub_windows(*this, container, minn, maxn); }
int Pattern::subngrams(std::vector<std::pair<Pattern, int>>& container, int minn, int maxn) const { return sub_windows(*this, container, minn, maxn); }
int Pattern::subngrams(std::vector<std::pair<PatternPointer, int>>& container, int minn, int maxn) const { return sub_windows(*this, container, minn, maxn); }

// ---------------------------------------------------------------------------------------------------
// gap masks (algorithms.h)
// ---------------------------------------------------------------------------------------------------
uint32_t vector2mask(const std::vector<std::pair<int, int>>& skips) {
    uint32_t mask = 0;
    for (const auto& gap : skips)
        for (int i = gap.first; i < gap.first + gap.second && i < 31; ++i) mask |= bitmask_of(i);
    return mask;
}
std::vector<std::pair<int, int>> mask2vector(const uint32_t mask, const int n) {
    std::vector<std::pair<int, int>> gaps;
    int                              k = 0;
    while (k < n) {
        if (!(mask & bitmask_of(k))) {
            ++k;
            continue;
        }
        const int begin = k;
        while (k < n && (mask & bitmask_of(k))) ++k;
        gaps.push_back(std::make_pair(begin, k - begin));
    }
    return gaps;
}
std::vector<std::pair<int, int>> get_consecutive_gaps(const int n, const int leftmargin, const int rightmargin) {
    std::vector<std::pair<int, int>> gaps;
    for (int begin = leftmargin; begin < n; ++begin)
        for (int length = (n - rightmargin) - begin; length > 0; --length) gaps.push_back(std::make_pair(begin, length));
    return gaps;
}
uint32_t reversemask(uint32_t mask, const unsigned int n) {
    const uint32_t inverted = ~mask;
    return n >= 32 ? 0u : (uint32_t)(inverted << n) >> n;  // the reference's arithmetic: n bits up, n bits down, in 32 bits
}
int maskheadskip(uint32_t mask, const unsigned int) {
    unsigned int i = 0;
    while (i < 31 && (mask & bitmask_of((int)i))) ++i;
    return (int)i;
}
int masktailskip(uint32_t mask, const unsigned int n) {
    unsigned int i = 0;
    while (i < n && (mask & bitmask_of((int)(n - i - 1)))) ++i;
    return (int)i;
}
std::vector<uint32_t> compute_skip_configurations(const int n, const int maxskips) {
    std::vector<uint32_t> masks;
    if (n < 3 || n > 32) return masks;  // (a gap mask is a uint32_t that covers neither end: beyond 32 tokens the reference's own loop is undefined)
    for (uint32_t inner = 1; inner < (uint32_t(1) << (n - 2)); ++inner) {
        const uint32_t mask = inner << 1;  // never the first or the last token
        if (n - 2 >= maxskips && mask2vector(mask, n).size() > (size_t)maxskips) continue;
        masks.push_back(mask);
    }
    return masks;
}

// ---------------------------------------------------------------------------------------------------
// class decoding
// ---------------------------------------------------------------------------------------------------
unsigned int bytestoint(const unsigned char* a, unsigned int* length) {
    unsigned int result = 0, i = 0;
    for (;; ++i) {
        const unsigned char b = a[i];
        if (i < 5) result |= (unsigned int)(b & 127) << (7 * i);
        if (b < 128) break;
    }
    if (length) *length = i + 1;
    return result;
}

unsigned char getdataversion(std::istream& in) {
    if (!in.good()) {
        std::cerr << "ERROR: Supplied data file can not be opened. Check whether it exists and whether you have proper permissions..." << std::endl;
        throw InternalError();
    }
    in.clear();
    in.seekg(0);
    unsigned char b = 0, version = 1;
    in.read((char*)&b, 1);
    if (b == 0xa2) {
        in.read((char*)&version, 1);
    } else {
        if (b > 5 && in.gcount() == 1) {
            std::cerr << "ERROR: Supplied data file is not a valid Colibri Data file, did you pass plain-text instead perhaps?..." << std::endl;
            throw InternalError();
        }
        in.clear();
        in.seekg(0);
    }
    return version;
}

ClassDecoder::ClassDecoder() : highestclass(0) {
    classes[unknownclass] = "{?}";
    classes[skipclass]    = "{*}";
    classes[flexclass]    = "{**}";
    classes[boundaryclass] = "{|}";
}
ClassDecoder::ClassDecoder(const std::string& filename) : ClassDecoder() { load(filename); }
void ClassDecoder::load(const std::string& filename) {
    std::ifstream in(filename);
    if (!in.good()) {
        std::cerr << "ERROR: Unable to load class file " << filename << std::endl;
        throw InternalError();
    }
    std::string line;
    while (std::getline(in, line)) {
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) continue;
        const unsigned int cls = (unsigned int)std::strtoul(line.substr(0, tab).c_str(), NULL, 10);
        classes[cls]           = line.substr(tab + 1);
        if (cls > highestclass) highestclass = cls;
    }
}
const std::string& ClassDecoder::operator[](unsigned int cls) const {
    static const std::string unknown = "{?}";
    auto                     it      = classes.find(cls);
    return it == classes.end() ? unknown : it->second;
}

// ---------------------------------------------------------------------------------------------------
// corpus files
// ---------------------------------------------------------------------------------------------------
namespace colibri_host {

// v1: token = length byte (1..127) + little-endian base-256 digits; 00 ends a sentence; 128/129 = skip/flex markers
static std::vector<unsigned char> v1_to_v2(const std::vector<unsigned char>& in) {
    std::vector<unsigned char> out;
    out.reserve(in.size());
    size_t i = 0;
    while (i < in.size()) {
        const unsigned char c = in[i];
        if (c == 0) {
            out.push_back(0);
            ++i;
        } else if (c < 128) {
            if (i + 1 + c > in.size()) {
                std::cerr << "ERROR: Invalid pattern data, unexpected end of file" << std::endl;
                throw InternalError();
            }
            uint32_t cls = 0;
            for (unsigned k = 0; k < c && k < 4; ++k) cls |= (uint32_t)in[i + 1 + k] << (8 * k);
            do {
                unsigned char b = cls & 127;
                cls >>= 7;
                if (cls) b |= 128;
                out.push_back(b);
            } while (cls);
            i += (size_t)c + 1;
        } else {
            if (c == 128) out.push_back(colibri_classes::skipclass);
            if (c == 129) out.push_back(colibri_classes::flexclass);
            ++i;
        }
    }
    return out;
}

std::vector<unsigned char> read_corpus_payload(std::istream& in) {
    const unsigned char version = getdataversion(in);  // leaves the stream after the 2-byte header for v2, at 0 for v1
    std::vector<unsigned char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (version >= 2) return raw;
    return v1_to_v2(raw);
}

Pattern read_model_pattern(std::istream& in, unsigned char classencodingversion) {
    std::vector<unsigned char> buf;
    unsigned char              c = 0;
    if (classencodingversion == 1) {
        while (in.read((char*)&c, 1)) {
            buf.push_back(c);
            if (c == 0) break;
            if (c < 128) {
                for (unsigned k = 0; k < c; ++k) {
                    unsigned char d = 0;
                    in.read((char*)&d, 1);
                    buf.push_back(d);
                }
            }
        }
        std::vector<unsigned char> v2 = v1_to_v2(buf);
        if (!v2.empty() && v2.back() == 0) v2.pop_back();
        return Pattern(v2.data(), v2.size());
    }
    bool prevhigh = false;
    while (in.read((char*)&c, 1)) {
        if (!prevhigh && c == 0) break;
        buf.push_back(c);
        prevhigh = c >= 128;
    }
    return Pattern(buf.data(), buf.size());
}

// ---------------------------------------------------------------------------------------------------
// driving the device library
// ---------------------------------------------------------------------------------------------------
namespace {
struct CtxGuard {
    colibri_ctx* c = nullptr;
    ~CtxGuard() { colibri_destroy(c); }
};
// One idle device context per GPU is kept between plain train() calls of a process: creating a context, reserving its working buffers in HBM and releasing them again
// cost ~15 ms of a ~34 ms call on a 10^8-token corpus (COLIBRI_HOST_TIMING=1), and a caller that trains model after model pays them every time. A context that ran
// with a constraint / filter / continuation set, or that failed, is never returned here. The cached context is not destroyed at exit (the HIP runtime may already be
// gone when static destructors run; the driver reclaims the memory). COLIBRI_CTX_CACHE=0 turns this off.
struct CtxCache {
    std::mutex                  m;
    std::map<int, colibri_ctx*> idle;
    static CtxCache& get() {
        static CtxCache* c = new CtxCache;  // (leaked on purpose, see above)
        return *c;
    }
    static bool on() {
        static const bool v = [] {
            const char* e = std::getenv("COLIBRI_CTX_CACHE");
            return !(e && e[0] == '0');
        }();
        return v;
    }
    colibri_ctx* take(int device) {
        if (!on()) return nullptr;
        std::lock_guard<std::mutex> l(m);
        auto                        it = idle.find(device);
        if (it == idle.end()) return nullptr;
        colibri_ctx* c = it->second;
        idle.erase(it);
        return c;
    }
    void evict(int device) {  // device < 0: every device's
        std::vector<colibri_ctx*> gone;
        {
            std::lock_guard<std::mutex> l(m);
            for (auto it = idle.begin(); it != idle.end();) {
                if (device < 0 || it->first == device) {
                    gone.push_back(it->second);
                    it = idle.erase(it);
                } else {
                    ++it;
                }
            }
        }
        for (colibri_ctx* c : gone) colibri_destroy(c);
    }
    void give(int device, colibri_ctx* c) {  // takes ownership
        if (c == nullptr) return;
        if (on()) {
            std::lock_guard<std::mutex> l(m);
            if (idle.find(device) == idle.end()) {
                idle[device] = c;
                return;
            }
        }
        colibri_destroy(c);
    }
};
// The arrays of the last released result (one set per process) wait for the next export: 150 MB of keys / offsets / counts for a 10^8-token model are ~37 000 fresh
// pages otherwise — 27-31 ms of page faults per train() once nothing else keeps the heap warm, twice the device work. They are handed over with their old size (no
// zero fill: the export overwrites every element). COLIBRI_RESULT_POOL=0 turns this off.
struct ResultPool {
    std::mutex                 m;
    bool                       full = false;
    std::vector<uint64_t>      key_off, ref_off;
    std::vector<unsigned char> key_bytes;
    std::vector<uint32_t>      counts, ref_sentence;
    std::vector<uint16_t>      ref_token;
    static ResultPool& get() {
        static ResultPool* p = new ResultPool;  // (leaked on purpose: released models may outlive static destructors)
        return *p;
    }
    static bool on() {
        static const bool v = [] {
            const char* e = std::getenv("COLIBRI_RESULT_POOL");
            return !(e && e[0] == '0');
        }();
        return v;
    }
};
void take_result_buffers(TrainResult& out) {
    if (!ResultPool::on()) return;
    ResultPool&                 p = ResultPool::get();
    std::lock_guard<std::mutex> l(p.m);
    if (!p.full) return;
    out.key_off.swap(p.key_off);
    out.key_bytes.swap(p.key_bytes);
    out.counts.swap(p.counts);
    out.ref_off.swap(p.ref_off);
    out.ref_sentence.swap(p.ref_sentence);
    out.ref_token.swap(p.ref_token);
    p.full = false;
}
[[noreturn]] void raise(colibri_ctx* c, int rc, const char* what) {
    std::cerr << "ERROR: " << what << " failed (status " << rc << "): " << (c ? colibri_last_error(c) : "no MI355X / HIP device available; there is no CPU fallback") << std::endl;
    throw InternalError();
}
}  // namespace

namespace {
void fetch_flexgrams(colibri_ctx* c, uint64_t nf, uint64_t kb, uint64_t nr, TrainResult& out) {
    out.key_off.assign(nf + 1, 0);
    out.key_bytes.assign(kb + 1, 0);
    out.counts.assign(nf + 1, 0);
    out.ref_off.assign(nf + 1, 0);
    out.ref_sentence.assign(nr + 1, 0);
    out.ref_token.assign(nr + 1, 0);
    const int rc = colibri_flexgrams_fetch(c, out.key_off.data(), out.key_bytes.data(), out.counts.data(), out.ref_off.data(), out.ref_sentence.data(), out.ref_token.data());
    if (rc != COLIBRI_OK) raise(c, rc, "colibri_flexgrams_fetch");
    out.counts.resize(nf);
}
}  // namespace

void device_flexgrams_resident(const std::shared_ptr<void>& device, TrainResult& out) {
    colibri_ctx* c  = static_cast<colibri_ctx*>(device.get());
    uint64_t     nf = 0, kb = 0, nr = 0;
    const int    rc = colibri_flexgrams_resident(c, &nf, &kb, &nr);
    if (rc != COLIBRI_OK) raise(c, rc, "colibri_flexgrams_resident");
    fetch_flexgrams(c, nf, kb, nr, out);
}

TrainResult::~TrainResult() {
    if (!ResultPool::on() || key_off.capacity() + key_bytes.capacity() + ref_sentence.capacity() < (1u << 18)) return;  // (small results are not worth keeping)
    ResultPool&                 p = ResultPool::get();
    std::lock_guard<std::mutex> l(p.m);
    if (p.full) return;
    p.key_off.swap(key_off);
    p.key_bytes.swap(key_bytes);
    p.counts.swap(counts);
    p.ref_off.swap(ref_off);
    p.ref_sentence.swap(ref_sentence);
    p.ref_token.swap(ref_token);
    p.full = true;
}

void device_train(const unsigned char* payload, uint64_t nbytes, const colibri_options& opt, uint32_t firstsentence, TrainResult& out, const ConstraintKeys* constraint,
                  bool keep_device, bool continuation, bool as_filter) {
    CtxGuard    g;
    const char* dev = std::getenv("COLIBRI_DEVICE");
    // COLIBRI_HOST_TIMING=1: where a call's wall time goes (context, upload, train, export), one line on stderr
    static const bool timing = std::getenv("COLIBRI_HOST_TIMING") != nullptr;
    typedef std::chrono::steady_clock clk;
    const auto  t0 = clk::now();
    auto        ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const int   device = dev ? std::atoi(dev) : 0;
    const bool  plain  = constraint == NULL && !keep_device;  // (the modes a set of keys switches on stay with a context: such a context is not shared)
    int         rc     = COLIBRI_OK;
    // The idle context of a plain train() keeps its working buffers in HBM (several GB at 10^8 tokens). A run that cannot use it — a constrained / kept-on-device one —
    // would have to find its own memory beside them: it gives the idle context up first. colibri_host::release_cached() does the same on request.
    if (plain) g.c = CtxCache::get().take(device);
    else CtxCache::get().evict(device);
    if (g.c == nullptr) rc = colibri_create(&g.c, device);
    if (rc != COLIBRI_OK) raise(nullptr, rc, "colibri_create");
    const auto t1 = clk::now();
    if ((rc = colibri_upload_corpus(g.c, payload, nbytes, firstsentence)) != COLIBRI_OK) raise(g.c, rc, "colibri_upload_corpus");
    const auto t2 = clk::now();
    if (constraint != NULL && as_filter) {
        const uint64_t             np   = constraint->off.empty() ? 0 : constraint->off.size() - 1;
        static const unsigned char none = 0;
        if (np && (rc = colibri_set_filter(g.c, constraint->off.data(), constraint->bytes.empty() ? &none : constraint->bytes.data(), np)) != COLIBRI_OK) raise(g.c, rc, "colibri_set_filter");
    } else if (constraint != NULL && continuation) {
        const uint64_t             np   = constraint->off.empty() ? 0 : constraint->off.size() - 1;
        static const unsigned char none = 0;
        if (np && (rc = colibri_set_continuation(g.c, constraint->off.data(), constraint->bytes.empty() ? &none : constraint->bytes.data(), np)) != COLIBRI_OK)
            raise(g.c, rc, "colibri_set_continuation");
    } else if (constraint != NULL) {
        const uint64_t np = constraint->off.empty() ? 0 : constraint->off.size() - 1;
        if (np == 0) {  // nothing can be a member: the model stays empty, the totals are still the corpus'
            colibri_options plain = opt;
            if (plain.mintokens < 2) plain.mintokens = 2;
            plain.minlength = 1;
            plain.maxlength = 1;
            if ((rc = colibri_train(g.c, &plain, &out.stats)) != COLIBRI_OK) raise(g.c, rc, "colibri_train");
            out.stats.npatterns = 0;
            out.stats.totaltypes = 0;
            out.stats.maxn = out.stats.minn = 0;  // an empty model: nothing of the plain order-1 run that only served the totals stays
            for (int n = 0; n < COLIBRI_MAX_ORDER; ++n) out.stats.found[n] = out.stats.kept[n] = out.stats.pruned[n] = out.stats.admitted[n] = 0;
            out.key_off.assign(1, 0);
            out.key_bytes.assign(1, 0);
            out.counts.clear();
            out.ref_off.assign(1, 0);
            return;
        }
        static const unsigned char none = 0;
        if ((rc = colibri_set_constraint(g.c, constraint->off.data(), constraint->bytes.empty() ? &none : constraint->bytes.data(), np)) != COLIBRI_OK) raise(g.c, rc, "colibri_set_constraint");
    }
    if ((rc = colibri_train(g.c, &opt, &out.stats)) != COLIBRI_OK) raise(g.c, rc, "colibri_train");
    const auto t3 = clk::now();
    uint64_t np = 0, kb = 0, nr = 0;
    if ((rc = colibri_result_sizes(g.c, &np, &kb, &nr)) != COLIBRI_OK) raise(g.c, rc, "colibri_result_sizes");
    const auto t4 = clk::now();
    take_result_buffers(out);  // (arrays of a released model, pages mapped; every element below is written by the export)
    out.key_off.resize(np + 1);
    out.key_bytes.resize(kb + 1);
    out.key_bytes[kb] = 0;
    out.counts.resize(np);
    if (!opt.indexed) {  // (an unindexed result has no reference arrays at all: value_from_result() asks ref_off.empty())
        out.ref_off.clear();
        out.ref_sentence.clear();
        out.ref_token.clear();
    }
    const auto t5 = clk::now();
    struct Report {
        bool on; clk::time_point t0, t1, t2, t3, t4, t5;
        ~Report() {
            if (!on) return;
            auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "COLIBRI_HOST_TIMING create %.2f upload %.2f train %.2f sizes %.2f alloc %.2f export %.2f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5),
                    ms(t5, clk::now()));
        }
    } report{timing, t0, t1, t2, t3, t4, t5};
    (void)ms;
    if (opt.indexed) {
        out.ref_off.resize(np + 1);
        out.ref_sentence.resize(nr + 1);
        out.ref_token.resize(nr + 1);
        out.ref_sentence[nr] = 0;
        out.ref_token[nr]    = 0;
        rc = colibri_export_indexed(g.c, out.key_off.data(), out.key_bytes.data(), out.counts.data(), out.ref_off.data(), out.ref_sentence.data(), out.ref_token.data());
        if (rc != COLIBRI_OK) raise(g.c, rc, "colibri_export_indexed");
        if (keep_device) {  // the caller may follow up on the resident model (computeflexgrams_fromskipgrams); released with the pending result
            out.device = std::shared_ptr<void>(g.c, [](void* p) { colibri_destroy(static_cast<colibri_ctx*>(p)); });
            g.c        = nullptr;
        }
    } else {
        uint32_t dummy = 0;
        rc = colibri_export_unindexed(g.c, out.key_off.data(), out.key_bytes.data(), np ? out.counts.data() : &dummy);
        if (rc != COLIBRI_OK) raise(g.c, rc, "colibri_export_unindexed");
    }
    if (plain && g.c != nullptr) {  // everything went well: the context waits for the next call
        CtxCache::get().give(device, g.c);
        g.c = nullptr;
    }
}

void release_cached() {
    CtxCache::get().evict(-1);
    if (ResultPool::on()) {
        ResultPool&                 p = ResultPool::get();
        std::lock_guard<std::mutex> l(p.m);
        std::vector<uint64_t>().swap(p.key_off);
        std::vector<unsigned char>().swap(p.key_bytes);
        std::vector<uint32_t>().swap(p.counts);
        std::vector<uint64_t>().swap(p.ref_off);
        std::vector<uint32_t>().swap(p.ref_sentence);
        std::vector<uint16_t>().swap(p.ref_token);
        p.full = false;
    }
}

void device_flexgrams(const std::vector<uint64_t>& key_off, const unsigned char* key_bytes, const std::vector<uint64_t>& ref_off, const uint32_t* ref_sentence,
                      const uint16_t* ref_token, TrainResult& out) {
    CtxGuard    g;
    const char* dev = std::getenv("COLIBRI_DEVICE");
    int         rc  = colibri_create(&g.c, dev ? std::atoi(dev) : 0);
    if (rc != COLIBRI_OK) raise(nullptr, rc, "colibri_create");
    const uint64_t np = key_off.empty() ? 0 : key_off.size() - 1;
    uint64_t       nf = 0, kb = 0, nr = 0;
    static const unsigned char none = 0;
    if ((rc = colibri_flexgrams(g.c, key_off.data(), key_bytes ? key_bytes : &none, ref_off.data(), ref_sentence, ref_token, np, &nf, &kb, &nr)) != COLIBRI_OK)
        raise(g.c, rc, "colibri_flexgrams");
    fetch_flexgrams(g.c, nf, kb, nr, out);
}

void drop_short_patterns(TrainResult& r, int minlength) {
    const size_t n = r.size();
    const bool   indexed = !r.ref_off.empty();
    size_t       w = 0;
    uint64_t     kb = 0, nr = 0;
    for (size_t j = 0; j < n; ++j) {
        const uint64_t a = r.key_off[j], b = r.key_off[j + 1];
        if ((int)token_count(r.key_bytes.data() + a, (size_t)(b - a)) < minlength) continue;
        std::memmove(r.key_bytes.data() + kb, r.key_bytes.data() + a, (size_t)(b - a));
        r.counts[w]  = r.counts[j];
        if (indexed) {
            const uint64_t ra = r.ref_off[j], rb = r.ref_off[j + 1];
            std::memmove(r.ref_sentence.data() + nr, r.ref_sentence.data() + ra, (size_t)(rb - ra) * sizeof(uint32_t));
            std::memmove(r.ref_token.data() + nr, r.ref_token.data() + ra, (size_t)(rb - ra) * sizeof(uint16_t));
            r.ref_off[w] = nr;
            nr += rb - ra;
        }
        r.key_off[w] = kb;
        kb += b - a;
        ++w;
    }
    r.key_off[w] = kb;
    r.key_off.resize(w + 1);
    r.counts.resize(w);
    if (indexed) {
        r.ref_off[w] = nr;
        r.ref_off.resize(w + 1);
    }
    r.stats.npatterns = w;
}

void print_training_log(const colibri_stats& s, const colibri_options& o, std::ostream& err) {
    if (o.dopatternperline) {  // one pass (reference include/patternmodel.h:1008-1009, :1196-1245)
        err << "Counting patterns from list, one per line" << std::endl;
        if (s.npatterns == 0) {
            err << "None found" << std::endl;
            return;
        }
        err << " Found " << s.npatterns << " ngrams... computing total word types prior to pruning..." << s.totaltypes << "...pruned 0...total kept: " << s.npatterns << std::endl;
        return;
    }
    for (int n = 1; n <= o.maxlength && n < COLIBRI_MAX_ORDER; ++n) {
        err << "Counting " << n << "-grams" << std::endl;
        if (s.found[n] == 0) {
            err << "None found" << std::endl;
            break;
        }
        err << " Found " << s.found[n] << " ngrams...";
        err << "pruned " << s.pruned[n] << "...total kept: " << s.kept[n] << std::endl;
    }
}

}  // namespace colibri_host

int getmodeltype(const std::string& filename) {
    std::ifstream in(filename, std::ios::in | std::ios::binary);
    unsigned char null = 1, type = 0;
    in.read((char*)&null, 1);
    in.read((char*)&type, 1);
    return (in.good() && null == 0) ? (int)type : 0;
}

// ---------------------------------------------------------------------------------------------------
// IndexedCorpus
// ---------------------------------------------------------------------------------------------------
void IndexedCorpus::load(std::istream& in, bool) {
    std::vector<unsigned char> payload = colibri_host::read_corpus_payload(in);
    delete[] corpus;
    corpussize = payload.size();
    corpus     = new unsigned char[corpussize + 16];
    std::memset(corpus, 0, corpussize + 16);
    if (corpussize) std::memcpy(corpus, payload.data(), corpussize);
    sentencestart.clear();
    delimiters         = 0;
    bool prevdelimiter = true, prevhigh = false;
    for (size_t i = 0; i < corpussize; ++i) {  // every delimiter opens a new sentence, empty ones included (reference src/pattern.cpp:1947-1958)
        if (prevdelimiter) {
            sentencestart.push_back(i);
            prevdelimiter = false;
        }
        if (!prevhigh && corpus[i] == 0) {
            prevdelimiter = true;
            ++delimiters;
        }
        prevhigh = corpus[i] >= 128;
    }
}
void IndexedCorpus::load(const std::string& filename, bool debug) {
    std::ifstream in(filename, std::ios::in | std::ios::binary);
    if (!in.good()) {
        std::cerr << "ERROR: Unable to load file " << filename << std::endl;
        throw InternalError();
    }
    load(in, debug);
}
PatternPointer IndexedCorpus::getsentence(int sentence) const {
    if (sentence < 1 || (size_t)sentence > sentencestart.size()) throw KeyError();
    const size_t b = sentencestart[sentence - 1];
    size_t       e = b;
    bool         prevhigh = false;
    while (e < corpussize && (prevhigh || corpus[e] != 0)) {
        prevhigh = corpus[e] >= 128;
        ++e;
    }
    return PatternPointer(corpus + b, e - b);
}
PatternPointer IndexedCorpus::getpattern(const IndexReference& begin, int length) const {
    const PatternPointer s = getsentence((int)begin.sentence);
    size_t               tok = 0, i = 0, startb = 0;
    bool                 found = begin.token == 0;
    for (; i < s.bytes && !found; ++i) {
        if (s.data[i] < 128) {
            ++tok;
            if (tok == begin.token) {
                startb = i + 1;
                found  = true;
            }
        }
    }
    if (!found) throw KeyError();
    size_t e = startb;
    int    got = 0;
    while (e < s.bytes && got < length) {
        if (s.data[e] < 128) ++got;
        ++e;
    }
    if (got < length) throw KeyError();
    return PatternPointer(s.data + startb, e - startb);
}
