// classencoder.cpp — host side of the class encoder (see include/classencoder.h). All corpus-proportional work goes through the
// C ABI (colibri_text_*); there is no CPU fallback.
#include "classencoder.h"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <numeric>

#include "colibri_hip.h"

namespace {
struct Ctx {
    colibri_ctx* c = nullptr;
    Ctx() {
        const char* dev = std::getenv("COLIBRI_DEVICE");
        const int   rc  = colibri_create(&c, dev ? std::atoi(dev) : 0);
        if (rc != COLIBRI_OK) {
            std::cerr << "ERROR: colibri_create failed (status " << rc << "): no MI355X / HIP device available; there is no CPU fallback" << std::endl;
            throw InternalError();
        }
    }
    ~Ctx() { colibri_destroy(c); }
    void check(int rc, const char* what) {
        if (rc == COLIBRI_OK) return;
        std::cerr << "ERROR: " << what << " failed (status " << rc << "): " << colibri_last_error(c) << std::endl;
        throw InternalError();
    }
};
bool ends_with(const std::string& s, const char* suffix) {
    const std::string x(suffix);
    return s.size() >= x.size() && s.compare(s.size() - x.size(), x.size(), x) == 0;
}
std::string read_text(const std::string& filename) {
    if (ends_with(filename, ".bz2") || ends_with(filename, ".xml")) {
        std::cerr << "ERROR: " << filename << ": bz2 / FoLiA input is not on the MI355X-accelerated path" << std::endl;
        throw InternalError();
    }
    std::ifstream in(filename, std::ios::in | std::ios::binary);
    if (!in) {
        std::cerr << "ERROR: File does not exist: " << filename << std::endl;
        throw InternalError();
    }
    return std::string((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
// the distinct words of `text` under `rules`, in the order of their first occurrence
struct Words {
    std::vector<uint32_t> start, length, count, order;  // order[k] = index of the k-th word to occur
};
void device_words(Ctx& g, const std::string& text, int rules, Words& w) {
    uint64_t nwords = 0, nd = 0;
    g.check(colibri_text_upload(g.c, (const uint8_t*)text.data(), text.size()), "colibri_text_upload");
    g.check(colibri_text_count(g.c, rules, &nwords, &nd), "colibri_text_count");
    w.start.resize(nd);
    w.length.resize(nd);
    w.count.resize(nd);
    g.check(colibri_text_words(g.c, w.start.data(), w.length.data(), w.count.data()), "colibri_text_words");
    w.order.resize(nd);
    std::iota(w.order.begin(), w.order.end(), 0u);
    std::sort(w.order.begin(), w.order.end(), [&](uint32_t a, uint32_t b) { return w.start[a] < w.start[b]; });
}
}  // namespace

ClassEncoder::ClassEncoder(const unsigned int minlength_, const unsigned int maxlength_) : highestclass(5), minlength(minlength_), maxlength(maxlength_) {}
ClassEncoder::ClassEncoder(const std::string& filename, const unsigned int minlength_, const unsigned int maxlength_) { load(filename, minlength_, maxlength_); }

void ClassEncoder::load(const std::string& filename, const unsigned int minlength_, const unsigned int maxlength_) {
    highestclass = 0;
    minlength    = minlength_;
    maxlength    = maxlength_;
    if (minlength || maxlength) {
        std::cerr << "ERROR: word length limits are not on the MI355X-accelerated path" << std::endl;
        throw InternalError();
    }
    std::ifstream IN(filename);
    if (!IN) {
        std::cerr << "ERROR: File does not exist: " << filename << std::endl;
        throw InternalError();
    }
    while (IN.good()) {
        std::string line;
        std::getline(IN, line);
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) continue;
        const unsigned int cls    = (unsigned int)std::atoi(line.substr(0, tab).c_str());
        classes[line.substr(tab + 1)] = cls;
        if (cls > highestclass) highestclass = cls;
    }
    classes["{?}"]  = unknownclass;
    classes["{*}"]  = skipclass;
    classes["{**}"] = flexclass;
    classes["{|}"]  = boundaryclass;
}

void ClassEncoder::processcorpus(const std::string& filename, std::unordered_map<std::string, unsigned int>& freqlist, std::unordered_set<std::string>* vocab) {
    if (vocab != NULL && !vocab->empty()) {
        std::cerr << "ERROR: vocabulary files are not on the MI355X-accelerated path" << std::endl;
        throw InternalError();
    }
    if (minlength || maxlength) {
        std::cerr << "ERROR: word length limits are not on the MI355X-accelerated path" << std::endl;
        throw InternalError();
    }
    const std::string text = read_text(filename);
    Ctx               g;
    Words             w;
    device_words(g, text, 0, w);
    for (const uint32_t k : w.order) freqlist[text.substr(w.start[k], w.length[k])] += w.count[k];  // a new word enters the map at its first occurrence, as freqlist[word]++ does
}

void ClassEncoder::buildclasses(const std::unordered_map<std::string, unsigned int>& freqlist, unsigned int threshold) {
    std::multimap<const unsigned int, const std::string> byfreq;  // ascending in (0 - freq) = descending frequency; equal keys keep their insertion order
    for (const auto& kv : freqlist)
        if (kv.second >= threshold) byfreq.insert(std::make_pair(0u - kv.second, kv.first));
    unsigned int cls = highestclass;
    for (const auto& kv : byfreq)
        if (!classes.count(kv.second)) classes[kv.second] = ++cls;
    highestclass = cls;
}

void ClassEncoder::build(const std::string& filename, unsigned int threshold, const std::string& vocabfile) {
    std::vector<std::string> files{filename};
    build(files, true, threshold, vocabfile);
}
void ClassEncoder::build(const std::vector<std::string>& files, bool quiet, unsigned int threshold, const std::string& vocabfile) {
    if (!vocabfile.empty()) {
        std::cerr << "ERROR: vocabulary files are not on the MI355X-accelerated path" << std::endl;
        throw InternalError();
    }
    std::unordered_map<std::string, unsigned int> freqlist;
    for (const auto& filename : files) {
        if (!quiet) std::cerr << "Processing " << filename << std::endl;
        processcorpus(filename, freqlist);
    }
    buildclasses(freqlist, threshold);
}

void ClassEncoder::save(const std::string& filename) {
    std::ofstream OUT(filename);
    for (const auto& kv : classes)
        if (kv.second != unknownclass) OUT << kv.second << '\t' << kv.first << std::endl;
}

void ClassEncoder::encodefile(const std::string& inputfilename, const std::string& outputfilename, bool allowunknown, bool autoaddunknown, bool append, bool ignorenewlines,
                              bool quiet) {
    if (ignorenewlines) {
        std::cerr << "ERROR: ignoring newlines (-n) is not on the MI355X-accelerated path" << std::endl;
        throw InternalError();
    }
    const std::string text = read_text(inputfilename);
    Ctx               g;
    Words             w;
    device_words(g, text, 1, w);
    // the reference stops at the last '\n' (a final unterminated line is never encoded): words that only occur after it play no part
    const size_t          lastnl = text.rfind('\n');
    const uint32_t        limit  = lastnl == std::string::npos ? 0u : (uint32_t)(lastnl + 1);
    std::vector<uint32_t> cls(w.start.size(), 0), repeat(w.start.size(), 0);
    for (const uint32_t k : w.order) {  // first-occurrence order: the order in which the reference meets (and, with -e, numbers) unknown words
        if (w.start[k] >= limit) continue;
        const std::string word = text.substr(w.start[k], w.length[k]);
        repeat[k]              = 1;
        if (word == "{*}") {
            cls[k] = skipclass;
        } else if (word == "{**}") {
            cls[k] = flexclass;
        } else if (word == "{?}") {
            cls[k] = unknownclass;
        } else if (word.substr(0, 2) == "{*" && word.substr(word.size() - 2, 2) == "*}") {
            const int n = std::atoi(word.substr(2, word.size() - 4).c_str());
            cls[k]      = skipclass;
            repeat[k]   = n > 0 ? (uint32_t)n : 0u;
        } else {
            auto it = classes.find(word);
            if (it != classes.end()) {
                cls[k] = it->second;
            } else if (autoaddunknown) {
                cls[k]        = ++highestclass;
                classes[word] = cls[k];
            } else if (!allowunknown) {
                throw UnknownTokenError();
            } else {
                cls[k] = unknownclass;
            }
        }
    }
    uint64_t outbytes = 0, ntokens = 0, nlines = 0;
    g.check(colibri_text_encode(g.c, cls.data(), repeat.data(), &outbytes, &ntokens, &nlines), "colibri_text_encode");
    std::vector<uint8_t> payload(outbytes);
    g.check(colibri_text_fetch(g.c, payload.data()), "colibri_text_fetch");
    std::ofstream OUT(outputfilename, append ? (std::ios::out | std::ios::binary | std::ios::app) : (std::ios::out | std::ios::binary));
    if (!append) {
        const unsigned char header[2] = {0xa2, 2};
        OUT.write((const char*)header, 2);
    }
    OUT.write((const char*)payload.data(), (std::streamsize)payload.size());
    if (!quiet) std::cerr << "Encoded " << nlines << " lines" << std::endl;
}
