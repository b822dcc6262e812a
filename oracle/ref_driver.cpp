// oracle/ref_driver.cpp — TEST INFRASTRUCTURE, not product code.
//
// A small driver written for this repository that links against the *reference* library objects
// (built by oracle/Makefile from the sources where they lie under /root/reference, into
// oracle/_ref/) and calls the reference's public API for the hot path:
//   IndexedCorpus(filename)                                  reference include/patternstore.h:75
//   PatternModel<uint32_t>::train(filename, options)         reference include/patternmodel.h:1353
//   IndexedPatternModel<>::train(filename, options)          reference include/patternmodel.h:2828
//   SpookyHash::Hash64                                       reference include/SpookyV2.h:59
//   ClassEncoder::build / encodefile / save                  reference include/classencoder.h:93,174,224
// It prints results in the canonical text form the parity tests compare on (sorted by key bytes):
//   #tokens <u64>\n#types <u64>\n#patterns <u64>\n  then  <hex key bytes>\t<count>[\t<sentence>:<token> ...]\n
// It exists to (1) pin oracle/colibri_oracle.c against the real reference, (2) generate the golden
// fixtures under tests/golden/, (3) serve as the "reference" CPU baseline timed by bench.py.
// Nothing under colibri-core_amd/ may link or execute it.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "SpookyV2.h"
#include "classencoder.h"
#include "patternmodel.h"

namespace {

std::string hexof(const unsigned char* p, size_t n) {
    static const char* digits = "0123456789abcdef";
    std::string s;
    s.reserve(n * 2);
    for (size_t i = 0; i < n; ++i) {
        s.push_back(digits[p[i] >> 4]);
        s.push_back(digits[p[i] & 15]);
    }
    return s;
}

struct Row {
    std::string key;   // raw key bytes
    std::string rest;  // "\tcount[\trefs]"
};

template <class Model>
void collect_unindexed(Model& model, std::vector<Row>& rows) {
    for (auto it = model.begin(); it != model.end(); ++it) {
        const Pattern& p = it->first;
        Row r;
        r.key.assign(reinterpret_cast<const char*>(p.data), p.bytesize());
        r.rest = "\t" + std::to_string(model.occurrencecount(p));
        rows.push_back(std::move(r));
    }
}

void collect_indexed(IndexedPatternModel<>& model, std::vector<Row>& rows, bool sortrefs = false) {
    for (auto it = model.begin(); it != model.end(); ++it) {
        const Pattern& p = it->first;
        Row r;
        r.key.assign(reinterpret_cast<const char*>(p.data), p.bytesize());
        std::ostringstream os;
        os << "\t" << it->second.count() << "\t";
        bool first = true;
        if (sortrefs) it->second.sort();  // flexgram references arrive in map iteration order: canonical form = ascending, duplicates kept
        for (auto ref = it->second.begin(); ref != it->second.end(); ++ref) {
            if (!first) os << ' ';
            os << ref->sentence << ':' << ref->token;
            first = false;
        }
        r.rest = os.str();
        rows.push_back(std::move(r));
    }
}

void dump(std::ostream& out, uint64_t tokens, uint64_t types, std::vector<Row>& rows) {
    std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.key < b.key; });
    out << "#tokens " << tokens << "\n#types " << types << "\n#patterns " << rows.size() << "\n";
    for (const Row& r : rows) {
        out << hexof(reinterpret_cast<const unsigned char*>(r.key.data()), r.key.size()) << r.rest << "\n";
    }
}

int usage() {
    std::cerr << "usage:\n"
                 "  ref_driver train <corpus.colibri.dat> <mode:u|U|us|i|is> <maxlength> <mintokens>\n"
                 "             [-T minskiptypes] [-y mintokens_skipgrams] [-o model.out] [-d dump.txt] [-q] [-j constraintmodel] [-m minlength] [-F]\n"
                 "      u  = unindexed, streaming from file (patternmodeller -u)\n"
                 "      U  = unindexed, corpus preloaded in an IndexedCorpus (benchmarks.cpp test 5)\n"
                 "      us = unindexed + exhaustive skipgrams (preloaded corpus)\n"
                 "      i  = indexed (preloaded corpus);  is = indexed + skipgrams\n"
                 "      i2 / is2 = the same through the two-stage build of patternmodeller -2 (unindexed stage 1, constrained in-place stage 2)\n"
                 "  ref_driver load <model.colibri.patternmodel> <u|i> <dump.txt>\n"
                 "  ref_driver encode <text> <outprefix> [-t threshold] [-c classfile] [-e] [-U]\n"
                 "  ref_driver view <model> <u|i> <print|report|simplereport|histogram|info> <classfile>\n"
                 "  ref_driver relations <corpus> <classfile> <maxlength> <mintokens> <minskiptypes> <skipcontent|instances|templates> <out.txt>\n"
                 "  ref_driver hash <hex> [<hex> ...]\n"
                 "  ref_driver masks <n> <maxskips>\n";
    return 2;
}

std::vector<unsigned char> unhex(const std::string& s) {
    std::vector<unsigned char> v;
    for (size_t i = 0; i + 1 < s.size(); i += 2) v.push_back((unsigned char)strtol(s.substr(i, 2).c_str(), nullptr, 16));
    return v;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) return usage();
    const std::string cmd = argv[1];

    if (cmd == "hash") {
        for (int i = 2; i < argc; ++i) {
            std::vector<unsigned char> v = unhex(argv[i]);
            const uint64_t h = SpookyHash::Hash64(v.data(), v.size());
            char buf[32];
            snprintf(buf, sizeof buf, "%016llx", (unsigned long long)h);
            std::cout << argv[i] << "\t" << buf << "\n";
        }
        return 0;
    }

    if (cmd == "masks") {
        if (argc < 4) return usage();
        std::vector<uint32_t> masks = compute_skip_configurations(atoi(argv[2]), atoi(argv[3]));
        for (uint32_t m : masks) std::cout << m << "\n";
        return 0;
    }

    if (cmd == "encode") {  // the REFERENCE's class encoder, as colibri-classencode drives it (src/classencode.cpp:134-198)
        // ref_driver encode <text> <outprefix> [-t threshold] [-c classfile] [-e] [-U]      exit code 4 = unknown token
        if (argc < 4) return usage();
        const std::string text = argv[2], prefix = argv[3];
        unsigned int threshold = 0;
        std::string  classfile;
        bool         extend = false, allowunknown = false;
        for (int i = 4; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "-t" && i + 1 < argc) threshold = (unsigned int)atoi(argv[++i]);
            else if (a == "-c" && i + 1 < argc) classfile = argv[++i];
            else if (a == "-e") extend = true;
            else if (a == "-U") allowunknown = true;
        }
        std::vector<std::string> files{text};
        ClassEncoder             enc;
        if (!classfile.empty()) {
            enc = ClassEncoder(classfile);
            if (extend) enc.build(files, true, threshold, "");
        } else {
            enc.build(files, true, threshold, "");
        }
        try {
            enc.encodefile(text, prefix + ".colibri.dat", allowunknown, extend, false, false, true);
        } catch (const UnknownTokenError&) {
            return 4;
        }
        enc.save(prefix + ".colibri.cls");
        return 0;
    }

    if (cmd == "load") {  // the REFERENCE reads a .colibri.patternmodel (e.g. one written by the MI355X build) and dumps it canonically
        if (argc < 5) return usage();
        const std::string modelfile = argv[2], kind = argv[3], dumpout = argv[4];
        PatternModelOptions options;
        options.QUIET     = true;
        options.MINTOKENS = 1;
        std::vector<Row> rows;
        uint64_t tokens = 0, types = 0;
        if (kind == "u") {
            PatternModel<uint32_t> model(modelfile, options);
            tokens = model.tokens();
            types  = model.types();
            collect_unindexed(model, rows);
        } else {
            IndexedPatternModel<> model(modelfile, options);
            tokens = model.tokens();
            types  = model.types();
            collect_indexed(model, rows);
        }
        std::ofstream out(dumpout);
        dump(out, tokens, types, rows);
        return 0;
    }

    if (cmd == "view") {  // the REFERENCE's own print / report / histogram of a model file (golden text for the C++ face and the CLI)
        if (argc < 6) return usage();
        const std::string modelfile = argv[2], kind = argv[3], what = argv[4], classfile = argv[5];
        PatternModelOptions options;
        options.QUIET     = true;
        options.MINTOKENS = 1;
        ClassDecoder decoder(classfile);
        auto run = [&](auto& model) {
            if (what == "print") model.print(std::cout, decoder);
            else if (what == "report") model.report(std::cout, false);
            else if (what == "simplereport") model.report(std::cout, true);
            else if (what == "histogram") model.histogram(std::cout);
            else if (what == "info") model.info(std::cout);
        };
        if (kind == "u") {
            PatternModel<uint32_t> model(modelfile, options);
            run(model);
        } else {
            IndexedPatternModel<> model(modelfile, options);
            run(model);
        }
        return 0;
    }

    if (cmd == "relations") {  // ref_driver relations <corpus> <classfile> <maxlength> <mintokens> <minskiptypes> <skipcontent|instances|templates> <out.txt>
        // = colibri-patternmodeller -f corpus -c classfile -s -l .. -t .. -T .. --<filter>: every pattern of the model, then its relations
        // (src/patternmodeller.cpp:274-285, include/patternmodel.h:3595-3662)
        if (argc < 9) return usage();
        PatternModelOptions options;
        options.MAXLENGTH    = atoi(argv[4]);
        options.MINTOKENS    = atoi(argv[5]);
        options.MINSKIPTYPES = atoi(argv[6]);
        options.DOSKIPGRAMS  = true;
        options.QUIET        = true;
        const std::string filter = argv[7];
        ClassDecoder          decoder(argv[3]);
        IndexedCorpus         corpus(argv[2]);
        IndexedPatternModel<> model(&corpus);
        model.train(std::string(argv[2]), options);
        std::ofstream out(argv[8]);
        bool          first = true;
        for (auto it = model.begin(); it != model.end(); ++it) {
            const PatternPointer pp(it->first);
            out << pp.tostring(decoder) << std::endl;
            if (filter == "instances_api") {  // the C++ API proper (getinstances(const Pattern&), as src/test.cpp:1457-1460 uses it): through
                // outputrelations(PatternPointer, ...) the reference resolves to the PatternPointer overloads of the base class, which return nothing
                t_relationmap rel = model.getinstances(it->first);
                model.outputrelations(pp, rel, decoder, out, "INSTANCE-OF");
            } else if (filter == "templates_api") {
                t_relationmap rel = model.gettemplates(it->first);
                model.outputrelations(pp, rel, decoder, out, "TEMPLATE-OF");
            } else {
                model.outputrelations(pp, decoder, out, filter, first);
            }
            first = false;
        }
        return 0;
    }
    if (cmd != "train" || argc < 6) return usage();
    const std::string corpusfile = argv[2];
    const std::string mode       = argv[3];
    PatternModelOptions options;
    options.MAXLENGTH = atoi(argv[4]);
    options.MINTOKENS = atoi(argv[5]);
    options.QUIET     = false;
    std::string modelout, dumpout, constraintfile, inplacemodel, continuemodel, filterfile;
    bool flexfromskip = false;
    for (int i = 6; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-T" && i + 1 < argc) options.MINSKIPTYPES = atoi(argv[++i]);
        else if (a == "-y" && i + 1 < argc) options.MINTOKENS_SKIPGRAMS = atoi(argv[++i]);
        else if (a == "-o" && i + 1 < argc) modelout = argv[++i];
        else if (a == "-d" && i + 1 < argc) dumpout = argv[++i];
        else if (a == "-q") options.QUIET = true;
        else if (a == "-j" && i + 1 < argc) constraintfile = argv[++i];  // constrain by this model (patternmodeller -j, loaded as a PatternSetModel as src/patternmodeller.cpp:712-718 does)
        else if (a == "-m" && i + 1 < argc) options.MINLENGTH = atoi(argv[++i]);
        else if (a == "-W" && i + 1 < argc) options.MINTOKENS_UNIGRAMS = atoi(argv[++i]);
        else if (a == "-p" && i + 1 < argc) options.PRUNENONSUBSUMED = atoi(argv[++i]);
        else if (a == "-S" && i + 1 < argc) options.PRUNESUBSUMED = atoi(argv[++i]);
        else if (a == "-b" && i + 1 < argc) options.MAXBACKOFFLENGTH = atoi(argv[++i]);
        else if (a == "-L") { options.DOPATTERNPERLINE = true; options.MINTOKENS = 1; }  // patternmodeller -L: one pattern per line, implies -t 1 (src/patternmodeller.cpp:571-574, :677-678)
        else if (a == "-F") flexfromskip = true;  // computeflexgrams_fromskipgrams after training (patternmodeller -F S, src/patternmodeller.cpp:790-794), mode is
        else if (a == "-f" && i + 1 < argc) filterfile = argv[++i];  // train(..., filter): the patterns of this model file as a PatternSet<> (modes u and i)
        else if (a == "-E" && i + 1 < argc) continuemodel = argv[++i];  // continue training on this loaded model (train(..., continued = true): patternmodeller -i <model> -E), modes u and i
        else if (a == "-I" && i + 1 < argc) inplacemodel = argv[++i];  // constrained in-place rebuild of this model (patternmodeller -I -i <model>), modes u and i
        else return usage();
    }

    PatternSetModel* constrain = NULL;
    if (!constraintfile.empty()) {
        PatternModelOptions co = PatternModelOptions(options);
        co.DOREMOVEINDEX       = false;  // exactly what colibri-patternmodeller -j does: the set is loaded under the run's own thresholds
        constrain              = new PatternSetModel(constraintfile, co);
    }
    std::vector<Row> rows;
    uint64_t tokens = 0, types = 0;
    double load_s = 0, train_s = 0;
    using clk = std::chrono::steady_clock;

    if (!filterfile.empty() && (mode == "u" || mode == "i")) {  // what the Python binding does with its `filter` argument (include/patternmodel.h:880, :899-914)
        PatternModelOptions lo;
        lo.MINTOKENS = 1;
        lo.DOSKIPGRAMS = true;
        lo.QUIET = true;
        PatternSetModel source(filterfile, lo);
        PatternSet<>    filter;
        for (auto it = source.begin(); it != source.end(); ++it) filter.insert(*it);
        if (mode == "u") {
            PatternModel<uint32_t> model;
            model.train(corpusfile, options, NULL, &filter);
            tokens = model.tokens();
            types  = model.types();
            if (!modelout.empty()) model.write(modelout);
            if (!dumpout.empty()) collect_unindexed(model, rows);
        } else {
            IndexedCorpus         corpus(corpusfile);
            IndexedPatternModel<> model(&corpus);
            model.train(corpusfile, options, NULL, &filter);
            tokens = model.tokens();
            types  = model.types();
            if (!modelout.empty()) model.write(modelout);
            if (!dumpout.empty()) collect_indexed(model, rows);
        }
    } else if (!continuemodel.empty() && (mode == "u" || mode == "i")) {  // src/patternmodeller.cpp:350-358 with continued = true
        if (mode == "u") {
            PatternModel<uint32_t> model(continuemodel, options, NULL, NULL);
            model.train(corpusfile, options, NULL, NULL, true);
            tokens = model.tokens();
            types  = model.types();
            if (!modelout.empty()) model.write(modelout);
            if (!dumpout.empty()) collect_unindexed(model, rows);
        } else {
            IndexedCorpus         corpus(corpusfile);
            IndexedPatternModel<> model(continuemodel, options, NULL, &corpus);
            model.train(corpusfile, options, NULL, NULL, true);
            tokens = model.tokens();
            types  = model.types();
            if (!modelout.empty()) model.write(modelout);
            if (!dumpout.empty()) collect_indexed(model, rows);
        }
    } else if (!inplacemodel.empty() && (mode == "u" || mode == "i")) {  // src/patternmodeller.cpp:756-831
        PatternModelOptions optionscopy = PatternModelOptions(options);
        optionscopy.DORESET             = true;
        if (mode == "u") {
            PatternModel<uint32_t> model(inplacemodel, optionscopy, NULL, NULL);
            if (model.maxlength() > options.MAXLENGTH) options.MAXLENGTH = model.maxlength();
            if (model.minlength() < options.MINLENGTH) options.MINLENGTH = model.minlength();
            model.train(corpusfile, options, model.getinterface());
            tokens = model.tokens();
            types  = model.types();
            if (!modelout.empty()) model.write(modelout);
            if (!dumpout.empty()) collect_unindexed(model, rows);
        } else {
            IndexedCorpus         corpus(corpusfile);
            IndexedPatternModel<> model(inplacemodel, optionscopy, NULL, &corpus);
            if (model.maxlength() > options.MAXLENGTH) options.MAXLENGTH = model.maxlength();
            if (model.minlength() < options.MINLENGTH) options.MINLENGTH = model.minlength();
            model.train(corpusfile, options, model.getinterface());
            tokens = model.tokens();
            types  = model.types();
            if (!modelout.empty()) model.write(modelout);
            if (!dumpout.empty()) collect_indexed(model, rows);
        }
    } else if (mode == "u") {
        PatternModel<uint32_t> model;
        auto t0 = clk::now();
        model.train(corpusfile, options, constrain);
        train_s = std::chrono::duration<double>(clk::now() - t0).count();
        tokens = model.tokens();
        types  = model.types();
        if (!modelout.empty()) model.write(modelout);
        if (!dumpout.empty()) collect_unindexed(model, rows);
    } else if (mode == "U" || mode == "us") {
        auto t0 = clk::now();
        IndexedCorpus corpus(corpusfile);
        load_s = std::chrono::duration<double>(clk::now() - t0).count();
        if (mode == "us") options.DOSKIPGRAMS_EXHAUSTIVE = true;
        PatternModel<uint32_t> model(&corpus);
        t0 = clk::now();
        model.train(corpusfile, options, constrain);
        train_s = std::chrono::duration<double>(clk::now() - t0).count();
        tokens = model.tokens();
        types  = model.types();
        if (!modelout.empty()) model.write(modelout);
        if (!dumpout.empty()) collect_unindexed(model, rows);
    } else if (mode == "i" || mode == "is") {
        auto t0 = clk::now();
        IndexedCorpus corpus(corpusfile);
        load_s = std::chrono::duration<double>(clk::now() - t0).count();
        if (mode == "is") options.DOSKIPGRAMS = true;
        IndexedPatternModel<> model(&corpus);
        t0 = clk::now();
        model.train(corpusfile, options, constrain);
        train_s = std::chrono::duration<double>(clk::now() - t0).count();
        if (flexfromskip) {
            const size_t before = model.size();
            uint64_t     skiprefs = 0;
            for (auto it = model.begin(); it != model.end(); ++it)
                if (it->first.category() == SKIPGRAM) skiprefs += it->second.count();
            t0 = clk::now();
            const int found = model.computeflexgrams_fromskipgrams();
            const double flex_s = std::chrono::duration<double>(clk::now() - t0).count();
            std::cerr << "flexgrams " << found << std::endl;
            std::cerr << "flexgram_timing {\"patterns_before\": " << before << ", \"skipgram_refs\": " << skiprefs << ", \"flexgrams\": " << found << ", \"seconds\": " << flex_s << "}" << std::endl;
        }
        tokens = model.tokens();
        types  = model.types();
        if (!modelout.empty()) model.write(modelout);
        if (!dumpout.empty()) collect_indexed(model, rows, flexfromskip);
    } else if (mode == "i2" || mode == "is2") {
        // two-stage build as colibri-patternmodeller -2 does it (reference src/patternmodeller.cpp:627-663, :756-831): stage 1 an
        // unindexed model written to <tmp>.stage1, stage 2 an indexed model loaded from it (DORESET) and rebuilt in place, constrained by itself
        const std::string stage1 = (modelout.empty() ? dumpout : modelout) + ".stage1";
        IndexedCorpus     corpus(corpusfile);
        {
            PatternModelOptions o1 = options;
            o1.DOSKIPGRAMS         = false;
            PatternModel<uint32_t> m1(&corpus);
            m1.train(corpusfile, o1);
            m1.write(stage1);
        }
        if (mode == "is2") options.DOSKIPGRAMS = true;
        PatternModelOptions optionscopy = PatternModelOptions(options);
        optionscopy.DORESET             = true;
        IndexedPatternModel<> model(stage1, optionscopy, NULL, &corpus);
        if (model.maxlength() > options.MAXLENGTH) options.MAXLENGTH = model.maxlength();
        if (model.minlength() < options.MINLENGTH) options.MINLENGTH = model.minlength();
        auto t0 = clk::now();
        model.train(corpusfile, options, model.getinterface());
        train_s = std::chrono::duration<double>(clk::now() - t0).count();
        tokens  = model.tokens();
        types   = model.types();
        if (!modelout.empty()) model.write(modelout);
        if (!dumpout.empty()) collect_indexed(model, rows);
        std::remove(stage1.c_str());
    } else {
        return usage();
    }

    if (!dumpout.empty()) {
        std::ofstream out(dumpout);
        dump(out, tokens, types, rows);
    }
    printf("{\"mode\": \"%s\", \"maxlength\": %d, \"mintokens\": %d, \"tokens\": %llu, \"types\": %llu, \"load_s\": %.6f, \"train_s\": %.6f}\n",
           mode.c_str(), options.MAXLENGTH, options.MINTOKENS, (unsigned long long)tokens, (unsigned long long)types, load_s, train_s);
    return 0;
}
