// host_selftest — exercises the C++ face the way the reference's callers use it (src/test.cpp:1211-1232, src/benchmarks.cpp:228-237):
//   host_selftest cpu  <hamlet.v2.colibri.dat> <hamlet.v1.colibri.patternmodel>   host-only checks (formats, key types; no GPU)
//   host_selftest gpu  <corpus.colibri.dat> <out.model> <u|i> <maxlength> <mintokens>   train on the GPU, write the model, print a summary
//   host_selftest bench <corpus.colibri.dat> <maxlength> <mintokens> [reps]              PatternModel<uint32_t>::train() end to end, timed (bench.py: cxx_face_train_ms)
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "patternmodel.h"

static int fails = 0;
#define CHECK(cond)                                                               \
    do {                                                                          \
        if (!(cond)) {                                                            \
            std::cerr << "FAILED: " #cond " (" << __FILE__ << ":" << __LINE__ << ")" << std::endl; \
            ++fails;                                                              \
        }                                                                         \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    if (mode == "cpu") {
        // SpookyHash known answers of the reference
        const unsigned char k1[] = {6}, k3[] = {6, 7, 8}, k5[] = {0x86, 0x01, 0x90, 0x4e, 0x07};
        CHECK(Pattern(k1, 1).hash() == 0x5d3553ac0aa134faULL);
        CHECK(Pattern(k3, 3).hash() == 0x6ee4e90e1c0b57c9ULL);
        CHECK(Pattern(k5, 5).hash() == 0x626a44955a1c64f0ULL);
        CHECK(Pattern().hash() == 0);
        CHECK(Pattern(k5, 5).n() == 3 && Pattern(k5, 5).bytesize() == 5);
        CHECK(Pattern(k3, 3) == Pattern(k3, 3) && Pattern(k3, 3) != Pattern(k1, 1));
        // skipgram materialisation: gapped tokens -> 03, multi-byte token dropped whole
        unsigned char  w[] = {0x86, 0x01, 0x90, 0x4e, 0x07};
        PatternPointer pp(w, 5, 0b010);
        const unsigned char want[] = {0x86, 0x01, 0x03, 0x07};
        CHECK(Pattern(pp) == Pattern(want, 4));
        CHECK(Pattern(pp).category() == SKIPGRAM && Pattern(k3, 3).category() == NGRAM);
        std::vector<std::pair<PatternPointer, int>> grams;
        CHECK(PatternPointer(w, 5).ngrams(grams, 2) == 2 && grams[1].second == 1 && grams[1].first.bytes == 3);
        if (argc >= 3) {
            IndexedCorpus corpus{std::string(argv[2])};
            CHECK(corpus.sentences() == 40);  // reference src/test.cpp:1549
            CHECK(corpus.size() == 354);
            CHECK(corpus.getsentence(1).n() > 0);
            CHECK(corpus.getpattern(IndexReference(1, 0), 1).n() == 1);
        }
        if (argc >= 4) {  // the reference's own golden model (model version 1, v1 class encoding)
            PatternModelOptions options;
            options.QUIET = true;
            PatternModel<uint32_t> model(std::string(argv[3]), options);
            CHECK(model.size() == 111 && model.tokens() == 354 && model.types() == 186);
            CHECK(model.maxlength() == 7 && model.minlength() == 1);
            CHECK(model.occurrencecount(Pattern(k1, 1)) == 27);
            std::stringstream ss;
            model.write(ss);  // rewritten as version 2
            PatternModel<uint32_t> again(&ss, options);
            CHECK(again.size() == 111 && again.tokens() == 354 && again.types() == 186);
            CHECK(again.occurrencecount(Pattern(k1, 1)) == 27);
        }
        std::cout << (fails ? "FAILED" : "OK") << std::endl;
        return fails ? 1 : 0;
    }
    if (mode == "relations" && argc >= 9) {  // relations <corpus> <classfile> <maxlength> <mintokens> <minskiptypes> <filter> <out>: the C++ API of the relation
        // queries on a freshly trained indexed skipgram model, pattern by pattern in model order
        try {
            PatternModelOptions options;
            options.MAXLENGTH    = std::atoi(argv[4]);
            options.MINTOKENS    = std::atoi(argv[5]);
            options.MINSKIPTYPES = std::atoi(argv[6]);
            options.DOSKIPGRAMS  = true;
            options.QUIET        = true;
            const std::string filter = argv[7];
            ClassDecoder          decoder;
            decoder.load(std::string(argv[3]));
            IndexedCorpus         corpus{std::string(argv[2])};
            IndexedPatternModel<> model(&corpus);
            model.train(std::string(argv[2]), options);
            std::ofstream out(argv[8]);
            bool          first = true;
            for (IndexedPatternModel<>::iterator it = model.begin(); it != model.end(); ++it) {
                out << it->first.tostring(decoder) << std::endl;
                if (filter == "instances_api") {
                    t_relationmap rel = model.getinstances(it->first);
                    model.outputrelations(it->first, rel, decoder, out, "INSTANCE-OF");
                } else if (filter == "templates_api") {
                    t_relationmap rel = model.gettemplates(it->first);
                    model.outputrelations(it->first, rel, decoder, out, "TEMPLATE-OF");
                } else {
                    model.outputrelations(it->first, decoder, out, filter, first);
                }
                first = false;
            }
            std::cout << "OK" << std::endl;
            return 0;
        } catch (const std::exception& e) {
            std::cout << "FAILED: " << e.what() << std::endl;
            return 1;
        }
    }
    if (mode == "lookup" && argc >= 6) {
        // lookup <corpus> <u|i> <maxlength> <mintokens>: has() / occurrencecount() on a model FRESH from the device (answered from its flat arrays through the look-up table,
        // host/include/patternmodel.h FlatIndex) against the same model turned into map nodes by iteration — every pattern, and for every pattern two keys that are not in it
        try {
            PatternModelOptions options;
            options.MAXLENGTH = std::atoi(argv[4]);
            options.MINTOKENS = std::atoi(argv[5]);
            options.QUIET     = true;
            IndexedCorpus corpus{std::string(argv[2])};
            size_t        checked = 0, absent = 0;
            auto          run     = [&](auto& fresh, auto& walked) {
                fresh.train(std::string(argv[2]), options);
                walked.train(std::string(argv[2]), options);
                for (auto it = walked.begin(); it != walked.end(); ++it) {  // (begin() builds the node map of `walked`; `fresh` stays flat)
                    const Pattern& p = it->first;
                    CHECK(fresh.has(p));
                    CHECK(fresh.occurrencecount(p) == walked.occurrencecount(p));
                    ++checked;
                    std::string longer((const char*)p.data, p.bytesize());
                    longer.push_back((char)0x7e);  // one more token, class 126
                    const Pattern q((const unsigned char*)longer.data(), longer.size());
                    CHECK(fresh.has(q) == walked.has(q));
                    CHECK(fresh.occurrencecount(q) == walked.occurrencecount(q));
                    std::string other = longer;
                    other[0]          = (char)(other[0] == 0x7d ? 0x7c : 0x7d);
                    const Pattern o((const unsigned char*)other.data(), other.size());
                    CHECK(fresh.has(o) == walked.has(o));
                    absent += !walked.has(q);
                }
                CHECK(fresh.size() == walked.size());
            };
            if (std::string(argv[3]) == "i") {
                IndexedPatternModel<> fresh(&corpus), walked(&corpus);
                run(fresh, walked);
            } else {
                PatternModel<uint32_t> fresh(&corpus), walked(&corpus);
                run(fresh, walked);
            }
            std::cout << (fails ? "FAILED" : "OK") << " " << checked << " " << absent << std::endl;
            return fails ? 1 : 0;
        } catch (const std::exception& e) {
            std::cout << "EXCEPTION " << e.what() << std::endl;
            return 1;
        }
    }
    if (mode == "bench" && argc >= 5) {
        // what a caller of the reference's API pays: PatternModel<uint32_t>::train() on a preloaded corpus (src/benchmarks.cpp:228-237) through this C++ face — a device
        // context, the upload, colibri_train, the export of keys and counts to host memory — and the first look-up (the pattern map is materialised lazily)
        typedef std::chrono::steady_clock clk;
        PatternModelOptions options;
        options.MAXLENGTH = std::atoi(argv[3]);
        options.MINTOKENS = std::atoi(argv[4]);
        options.QUIET     = true;
        const int reps    = argc >= 6 ? std::atoi(argv[5]) : 3;
        try {
            const auto    l0 = clk::now();
            IndexedCorpus corpus{std::string(argv[2])};
            const double  load_ms = std::chrono::duration<double, std::milli>(clk::now() - l0).count();
            std::cout << "{\"corpus_load_ms\": " << load_ms << ", \"runs\": [";
            for (int r = 0; r < reps; ++r) {
                PatternModel<uint32_t> model(&corpus);
                const auto             t0 = clk::now();
                model.train(std::string(argv[2]), options);
                const auto          t1 = clk::now();
                const unsigned char k1[] = {6};
                const unsigned int  c6 = model.occurrencecount(Pattern(k1, 1));
                const auto          t2 = clk::now();
                std::cout << (r ? ", " : "") << "{\"train_ms\": " << std::chrono::duration<double, std::milli>(t1 - t0).count()
                          << ", \"first_lookup_ms\": " << std::chrono::duration<double, std::milli>(t2 - t1).count() << ", \"patterns\": " << model.size() << ", \"count_of_class_6\": " << c6 << "}";
            }
            std::cout << "]}" << std::endl;
        } catch (const InternalError&) {
            return 1;
        }
        return 0;
    }
    if (mode == "gpu" && argc >= 7) {
        PatternModelOptions options;
        options.MAXLENGTH = std::atoi(argv[5]);
        options.MINTOKENS = std::atoi(argv[6]);
        options.QUIET     = false;
        for (int a = 7; a < argc; ++a) {  // optional: p<N> = PRUNENONSUBSUMED, S<N> = PRUNESUBSUMED, b<N> = MAXBACKOFFLENGTH
            if (argv[a][0] == 'b') options.MAXBACKOFFLENGTH = std::atoi(argv[a] + 1);
            if (argv[a][0] == 'p') options.PRUNENONSUBSUMED = std::atoi(argv[a] + 1);
            if (argv[a][0] == 'S') options.PRUNESUBSUMED = std::atoi(argv[a] + 1);
        }
        const std::string kind = argv[4];
        try {
            // f<model file>: train(..., filter) with the patterns of that file as the PatternSet<> (what the Python binding of the reference passes)
            PatternSet<> filter;
            for (int a = 7; a < argc; ++a) {
                if (argv[a][0] != 'f') continue;
                PatternModelOptions lo;
                lo.MINTOKENS   = 1;
                lo.DOSKIPGRAMS = true;
                lo.QUIET       = true;
                PatternSetModel source(std::string(argv[a] + 1), lo);
                for (PatternSetModel::iterator it = source.begin(); it != source.end(); ++it) filter.insert(it->first);
            }
            if (filter.size() && kind == "u") {
                PatternModel<uint32_t> model;
                model.train(std::string(argv[2]), options, NULL, &filter);
                model.write(std::string(argv[3]));
                std::cout << model.size() << " " << model.tokens() << " " << model.types() << " " << model.maxlength() << std::endl;
            } else if (filter.size()) {
                IndexedCorpus         corpus{std::string(argv[2])};
                IndexedPatternModel<> model(&corpus);
                model.train(std::string(argv[2]), options, NULL, &filter);
                model.write(std::string(argv[3]));
                std::cout << model.size() << " " << model.tokens() << " " << model.types() << " " << model.maxlength() << std::endl;
            } else if (kind == "u") {
                PatternModel<uint32_t> model;
                model.train(std::string(argv[2]), options);
                model.write(std::string(argv[3]));
                std::cout << model.size() << " " << model.tokens() << " " << model.types() << " " << model.maxlength() << std::endl;
            } else if (kind == "U") {  // preloaded corpus, as src/benchmarks.cpp test 5
                IndexedCorpus          corpus{std::string(argv[2])};
                PatternModel<uint32_t> model(&corpus);
                model.train(std::string(argv[2]), options);
                model.write(std::string(argv[3]));
                const unsigned char k1[] = {6};
                std::cout << model.size() << " " << model.tokens() << " " << model.types() << " " << model.maxlength() << " " << model.occurrencecount(Pattern(k1, 1)) << std::endl;
            } else {
                IndexedCorpus         corpus{std::string(argv[2])};
                IndexedPatternModel<> model(&corpus);
                if (kind == "is") options.DOSKIPGRAMS = true;
                model.train(std::string(argv[2]), options);
                model.write(std::string(argv[3]));
                std::cout << model.size() << " " << model.tokens() << " " << model.types() << " " << model.maxlength() << std::endl;
            }
        } catch (const std::exception& e) {
            std::cout << "EXCEPTION " << e.what() << std::endl;
            return 1;
        }
        return 0;
    }
    return 2;
}
