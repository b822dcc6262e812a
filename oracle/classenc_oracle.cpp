// oracle/classenc_oracle.cpp — TEST INFRASTRUCTURE, not product code.
// CPU restatement of the reference's class encoder: word frequency list -> classes -> .colibri.cls text and
// .colibri.dat (v2) bytes. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
// It is C++ (not C) on purpose: the class a word receives depends, among words of equal frequency, on the iteration order of the
// reference's std::unordered_map<std::string, unsigned> (filled in first-occurrence order) and of a std::multimap with equal
// keys — the same libstdc++ containers are used here so that ties fall the same way. Pinned against the real reference
// (oracle/_ref/ref_driver encode ...) by tests/test_classenc.py and the goldens under tests/golden/classenc/.
//   processcorpus   reference src/classencoder.cpp:156-188   (split each getline() line at ' ', right-trim " \t\n\r")
//   buildclasses    reference src/classencoder.cpp:213-229   (multimap by -freq, classes from highestclass+1)
//   save            reference src/classencoder.cpp:270-277
//   load            reference src/classencoder.cpp:94-132
//   encodestring    reference src/classencoder.cpp:369-436   (right-trim " \t\n\r\b", {*} {**} {?} {*N*}, unknown words)
//   encodefile      reference src/classencoder.cpp:550-600   (A2 02 header, one 00 per line, a last line without '\n' is dropped)
//   inttobytes      reference src/classencoder.cpp:22-42
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
// common.cpp:10-19 — despite its name this trims the END only
std::string rtrim(const std::string& t, const char* ws) {
    const size_t found = t.find_last_not_of(ws);
    return found == std::string::npos ? std::string() : t.substr(0, found + 1);
}
// the lines std::getline would hand out while the stream is good(): every '\n'-terminated line, then the unterminated rest
// (possibly empty). `terminated` tells whether the line ended in '\n' (encodefile drops the one that did not, :569-570).
struct Line {
    const char* p;
    size_t      n;
    bool        terminated;
};
std::vector<Line> lines_of(const char* text, size_t n) {
    std::vector<Line> out;
    size_t            start = 0;
    for (size_t i = 0; i < n; ++i)
        if (text[i] == '\n') {
            out.push_back({text + start, i - start, true});
            start = i + 1;
        }
    out.push_back({text + start, n - start, false});
    return out;
}
void append_varint(std::vector<unsigned char>& out, unsigned int cls) {  // classencoder.cpp:22-42
    while (cls >= 128u) {
        out.push_back((unsigned char)((cls & 127u) | 128u));
        cls >>= 7;
    }
    out.push_back((unsigned char)cls);
}
struct Encoder {
    std::unordered_map<std::string, unsigned int> classes;
    unsigned int                                  highestclass = 5;
    void processcorpus(const char* text, size_t n, std::unordered_map<std::string, unsigned int>& freqlist) const {
        for (const Line& ln : lines_of(text, n)) {
            const std::string line(ln.p, ln.n);
            int               start = 0;
            const int         s     = (int)line.size();
            for (int i = 0; i < s; i++) {
                if (line[i] == ' ' || i == s - 1) {
                    const int   offset = (i == s - 1) ? 1 : 0;
                    std::string word(line.begin() + start, line.begin() + i + offset);
                    if (word.length() > 0 && word != "\r" && word != "\t" && word != " ") {
                        word = rtrim(word, " \t\n\r");
                        freqlist[word]++;
                    }
                    start = i + 1;
                }
            }
        }
    }
    void buildclasses(const std::unordered_map<std::string, unsigned int>& freqlist, unsigned int threshold) {
        std::multimap<const unsigned int, const std::string> revfreqlist;
        for (auto const& kv : freqlist)
            if (kv.second >= threshold) revfreqlist.insert(std::make_pair(-1 * kv.second, kv.first));
        int cls = (int)highestclass;
        for (const auto& it : revfreqlist)
            if (!classes.count(it.second)) classes[it.second] = (unsigned int)++cls;
        highestclass = (unsigned int)cls;
    }
    void load(const char* cls, size_t n) {
        highestclass = 0;
        for (const Line& ln : lines_of(cls, n)) {
            const std::string line(ln.p, ln.n);
            for (size_t i = 0; i < line.size(); i++)
                if (line[i] == '\t') {
                    const unsigned int c = (unsigned int)atoi(line.substr(0, i).c_str());
                    classes[line.substr(i + 1)] = c;
                    if (c > highestclass) highestclass = c;
                    break;
                }
        }
        classes["{?}"]  = 2;
        classes["{*}"]  = 3;
        classes["{**}"] = 4;
        classes["{|}"]  = 1;
    }
    std::string save() const {
        std::string out;
        for (auto const& kv : classes)
            if (kv.second != 2) out += std::to_string(kv.second) + "\t" + kv.first + "\n";
        return out;
    }
    // returns false on an unknown word when neither allowunknown nor autoaddunknown is set (UnknownTokenError)
    bool encodestring(const std::string& line, std::vector<unsigned char>& out, bool allowunknown, bool autoaddunknown) {
        int       start = 0;
        const int l     = (int)line.length();
        for (int i = 0; i < l; i++) {
            if (line[i] == ' ' || i == l - 1) {
                std::string word = line[i] == ' ' ? std::string(line.begin() + start, line.begin() + i) : std::string(line.begin() + start, line.begin() + i + 1);
                word             = rtrim(word, " \t\n\r\b");
                start            = i + 1;
                if (word.length() > 0 && word != "\r" && word != "\t" && word != " ") {
                    unsigned int cls;
                    if (word == "{*}") {
                        out.push_back(3);
                        continue;
                    } else if (word == "{**}") {
                        out.push_back(4);
                        continue;
                    } else if (word == "{?}") {
                        out.push_back(2);
                        continue;
                    } else if (word.substr(0, 2) == "{*" && word.substr(word.size() - 2, 2) == "*}") {
                        const int skipcount = atoi(word.substr(2, word.size() - 4).c_str());
                        for (int j = 0; j < skipcount; j++) out.push_back(3);
                        continue;
                    } else if (classes.find(word) == classes.end()) {
                        if (autoaddunknown) {
                            cls           = ++highestclass;
                            classes[word] = cls;
                        } else if (!allowunknown) {
                            return false;
                        } else {
                            cls = 2;
                        }
                    } else {
                        cls = classes[word];
                    }
                    append_varint(out, cls);
                }
            }
        }
        return true;
    }
    bool encodefile(const char* text, size_t n, std::vector<unsigned char>& out, bool allowunknown, bool autoaddunknown, bool append) {
        if (!append) {
            out.push_back(0xa2);
            out.push_back(2);
        }
        for (const Line& ln : lines_of(text, n)) {
            if (!ln.terminated) break;  // getline hit EOF: the stream is no longer good() (:569-570)
            if (!encodestring(std::string(ln.p, ln.n), out, allowunknown, autoaddunknown)) return false;
            out.push_back(0);
        }
        return true;
    }
};
unsigned char* give(const void* p, size_t n) {
    unsigned char* b = (unsigned char*)malloc(n ? n : 1);
    if (n) memcpy(b, p, n);
    return b;
}
}  // namespace

extern "C" {
// What colibri-classencode does (reference src/classencode.cpp:134-198) for one corpus:
//   existing_cls == NULL:            build classes from the corpus (threshold), save, encode
//   existing_cls, extend == 0:       load, encode (unknown words: class 2 if allowunknown, else error -> returns 1)
//   existing_cls, extend != 0 (-e):  load, build (new words get classes above the highest), save, encode with autoaddunknown, save again if classes were added
// Outputs are malloc'ed (free with classenc_oracle_free). Returns 0 ok, 1 unknown token.
int classenc_oracle_run(const unsigned char* text, uint64_t n, unsigned int threshold, int allowunknown, const unsigned char* existing_cls, uint64_t cls_n, int extend,
                        unsigned char** cls_out, uint64_t* cls_out_n, unsigned char** dat_out, uint64_t* dat_out_n) {
    Encoder enc;
    if (existing_cls != NULL) enc.load((const char*)existing_cls, (size_t)cls_n);
    if (existing_cls == NULL || extend) {
        std::unordered_map<std::string, unsigned int> freqlist;
        enc.processcorpus((const char*)text, (size_t)n, freqlist);
        enc.buildclasses(freqlist, threshold);
    }
    std::vector<unsigned char> dat;
    const bool                 ok = enc.encodefile((const char*)text, (size_t)n, dat, allowunknown != 0, extend != 0, false);
    const std::string          cls = enc.save();
    *cls_out   = give(cls.data(), cls.size());
    *cls_out_n = cls.size();
    *dat_out   = give(dat.data(), dat.size());
    *dat_out_n = dat.size();
    return ok ? 0 : 1;
}
void classenc_oracle_free(void* p) { free(p); }
}
