// colibri-patternmodeller (MI355X build) — a drop-in for the reference's command-line driver on the accelerated path.
// Same flags and flag meanings as reference src/patternmodeller.cpp:404-858 for: -f -c -o -i -u -t -l -m -b -W -s -y -T -P -R -r -H -e -D -h
// (build a model from a .colibri.dat, save it, load a model, print / report / histogram), plus -2 (two-stage build), -p (prune by
// subsumption), -j (constrain by a model), -I (constrained in-place rebuild of the model given with -i) and -F S (flexgrams abstracted from the
// skipgrams of a freshly built indexed model) and -L (one pattern per line). Flags that select paths outside the accelerated subset (-E -M -Q -q -g, -F <npmi> ...) are
// reported and rejected instead of being silently ignored.
// All counting happens in libcolibri_hip.so; this file only parses options and calls the C++ face.
#include <getopt.h>

#include <cstdlib>
#include <iostream>
#include <string>

#include "patternmodel.h"

namespace {

void usage() {
    std::cerr << "colibri-patternmodeller (MI355X-native build of the Colibri Core pattern-model builder)\n"
                 "Syntax: colibri-patternmodeller [options]\n"
                 " Input/output:\n"
                 "\t-f|--datafile <file>        class-encoded corpus (.colibri.dat, v2 or v1)\n"
                 "\t-c|--classfile <file>       class file (.colibri.cls), needed to print patterns as text\n"
                 "\t-i|--inputmodel <file>      load a model (.colibri.patternmodel)\n"
                 "\t-o|--outputmodel <file>     write the model\n"
                 " Building (runs on the GPU):\n"
                 "\t-t|--threshold <n>          occurrence threshold (default 2)\n"
                 "\t-u|--unindexed              unindexed model (default is indexed)\n"
                 "\t-l|--maxlength <n>          maximum pattern length (default 100)\n"
                 "\t-m|--minlength <n>          minimum pattern length (default 1)\n"
                 "\t-b|--backofflength <n>      maximum back-off length (default 100)\n"
                 "\t-W|--wordthreshold <n>      secondary word occurrence threshold\n"
                 "\t-s|--skipgrams              compute skipgrams\n"
                 "\t-y|--skipthreshold <n>      occurrence threshold for skipgrams\n"
                 "\t-T|--skiptypes <n>          skip type threshold (default 2)\n"
                 "\t-e|--expand <n>             sentence offset given to the first sentence; with -i and -f: train on the loaded model (needs -E)\n"
                 "\t-E|--selfexpand             continued training: only the pattern lengths the loaded model lacks are counted (e.g. -i m -f corpus -e 1 -E -l 8)\n"
                 "\t-2|--twostage               two-stage build of an indexed model (needs -o): same result as the reference's -2\n"
                 "\t-p|--prune <n>              prune the (k-1)-grams that no k-gram of the model contains, from k = n downwards\n"
                 "\t-j|--constraints <file>     only count patterns that occur in this model (any threshold, any minimum length)\n"
                 "\t-L|--patternlist            the data file is a list of one pattern per line: no sub-n-grams, implies -t 1 and -u\n"
                 "\t-F|--flexgrams S            flexgrams by abstracting over skipgrams (implies -s); indexed models built from a corpus\n"
                 "\t-I|--constrained            in-place rebuild: recount the patterns of the model given with -i on the corpus given with -f\n"
                 "\t--gpus <n>                  train sentence-sharded across n GPUs of this node (RCCL; not with -j, -I, -L)\n"
                 "\t--skipcontent               after the views: every pattern, then the skip content of the skipgrams (needs -c and a corpus)\n"
                 "\t--instances | --templates   as in the reference, these print the patterns only (its relation getters are not reached from here)\n"
                 " Viewing:\n"
                 "\t-P|--print   -R|--report   -r|--simplereport   -H|--histogram\n"
                 "\t-D|--debug   -h|--help\n";
}

PatternSetModel* g_constraint = NULL;  // -j
bool             g_inplace    = false; // -I
bool             g_flexfromskip = false; // -F S
bool             g_continued  = false; // -E: train(..., continued = true) on the loaded model
bool             g_expand     = false; // -e: train on the loaded model at all (reference src/patternmodeller.cpp:356)
std::string      g_relations;            // --skipcontent / --instances / --templates

template <class ModelType>
int run(ModelType& model, const std::string& corpusfile, const std::string& inputmodel, const std::string& outputmodel, const PatternModelOptions& options_in, uint32_t firstsentence,
        bool doprint, bool doreport, bool nocoverage, bool dohistogram, const ClassDecoder* decoder) {
    PatternModelOptions options = options_in;
    if (g_inplace) {  // reference src/patternmodeller.cpp:756-831
        std::cerr << "Constrained in-place rebuild (--constrained|-I) enabled, on " << corpusfile << std::endl;
        model.load(inputmodel, options);
        std::cerr << "(" << model.size() << " patterns)" << std::endl;
        if (model.maxlength() > options.MAXLENGTH) options.MAXLENGTH = model.maxlength();
        if (model.minlength() < options.MINLENGTH) options.MINLENGTH = model.minlength();
        model.train(corpusfile, options, model.getinterface(), NULL, false, firstsentence);
    } else if (!inputmodel.empty()) {
        model.load(inputmodel, options);
        if (!corpusfile.empty() && g_expand) {  // reference src/patternmodeller.cpp:356-358; only the continued form (-E: add the orders the model lacks) runs here
            std::cerr << "Expanding model on  " << corpusfile << std::endl;
            model.train(corpusfile, options, g_constraint, NULL, g_continued, firstsentence);
        }
    } else {
        model.train(corpusfile, options, g_constraint, NULL, false, firstsentence);
        if (g_flexfromskip && options.DOSKIPGRAMS) {  // reference src/patternmodeller.cpp:337-341, :790-794 (messages as there, file name glued on)
            std::cerr << "Computing flexgrams from skipgrams" << corpusfile << std::endl;
            const int found = model.computeflexgrams_fromskipgrams();
            std::cerr << found << " flexgrams found" << corpusfile << std::endl;
        }
    }
    if (!outputmodel.empty()) {
        std::cerr << "Writing model to " << outputmodel << std::endl;
        model.write(outputmodel);
    }
    // the views, in the reference's order and on the same stream (viewmodel, reference src/patternmodeller.cpp:242-289)
    std::cerr << "Generating desired views..." << std::endl;
    if (doprint) {
        if (decoder == NULL) std::cerr << "ERROR: Unable to print model, no class file specified (--classfile)" << std::endl;
        else model.print(std::cout, *decoder);
    }
    if (doreport) model.report(std::cout, nocoverage);
    if (dohistogram) model.histogram(std::cout);
    if (!g_relations.empty()) {  // every pattern of the model, then its relations (reference src/patternmodeller.cpp:274-285)
        bool first = true;
        for (typename ModelType::iterator it = model.begin(); it != model.end(); ++it) {
            std::cout << it->first.tostring(*decoder) << std::endl;
            model.outputrelations(it->first, *decoder, std::cout, g_relations, first);
            first = false;
        }
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    std::string         corpusfile, classfile, inputmodel, outputmodel, constraintfile;
    PatternModelOptions options;
    bool                unindexed = false, doprint = false, doreport = false, nocoverage = false, dohistogram = false, twostage = false;
    uint32_t            firstsentence = 1;
    static struct option longopts[] = {{"datafile", required_argument, 0, 'f'},    {"classfile", required_argument, 0, 'c'},      {"inputmodel", required_argument, 0, 'i'},
                                       {"outputmodel", required_argument, 0, 'o'}, {"threshold", required_argument, 0, 't'},      {"unindexed", no_argument, 0, 'u'},
                                       {"maxlength", required_argument, 0, 'l'},   {"minlength", required_argument, 0, 'm'},      {"backofflength", required_argument, 0, 'b'},
                                       {"wordthreshold", required_argument, 0, 'W'}, {"skipgrams", no_argument, 0, 's'},          {"skipthreshold", required_argument, 0, 'y'},
                                       {"skiptypes", required_argument, 0, 'T'},   {"expand", required_argument, 0, 'e'},         {"print", no_argument, 0, 'P'},
                                       {"report", no_argument, 0, 'R'},            {"simplereport", no_argument, 0, 'r'},         {"histogram", no_argument, 0, 'H'},
                                       {"debug", no_argument, 0, 'D'},             {"help", no_argument, 0, 'h'},                 {"twostage", no_argument, 0, '2'},          {"constraints", required_argument, 0, 'j'},    {"constrained", no_argument, 0, 'I'},    {"flexgrams", required_argument, 0, 'F'},    {"patternlist", no_argument, 0, 'L'},
                                       {"skipcontent", no_argument, 0, 1001},      {"instances", no_argument, 0, 1002},           {"templates", no_argument, 0, 1003},
                                       {"gpus", required_argument, 0, 1004},         {"selfexpand", no_argument, 0, 'E'},
                                       {0, 0, 0, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "f:c:i:o:t:ul:m:b:W:sy:T:e:PRrHDh2j:Ip:EF:LMQq:gZV", longopts, NULL)) != -1) {
        switch (c) {
            case 1004: colibri_host::set_gpus(std::atoi(optarg)); break;
            case 'f': corpusfile = optarg; break;
            case 'c': classfile = optarg; break;
            case 'i': inputmodel = optarg; break;
            case 'o': outputmodel = optarg; break;
            case 't': options.MINTOKENS = std::atoi(optarg); break;
            case 'u': unindexed = true; break;
            case 'l': options.MAXLENGTH = std::atoi(optarg); break;
            case 'm': options.MINLENGTH = std::atoi(optarg); break;
            case 'b': options.MAXBACKOFFLENGTH = std::atoi(optarg); break;
            case 'W': options.MINTOKENS_UNIGRAMS = std::atoi(optarg); break;
            case 's': options.DOSKIPGRAMS = true; break;
            case 'y': options.MINTOKENS_SKIPGRAMS = std::atoi(optarg); break;
            case 'T': options.MINSKIPTYPES = std::atoi(optarg); break;
            case 'e':
                firstsentence = (uint32_t)std::atoi(optarg);
                g_expand      = true;
                break;
            case 'E': g_continued = true; break;
            case 'P': doprint = true; break;
            case 'R': doreport = true; break;
            case 'r':  // report without the coverage columns (reference :603-606)
                doreport   = true;
                nocoverage = true;
                break;
            case 'H': dohistogram = true; break;
            case 'D': options.DEBUG = true; break;
            case '2': twostage = true; break;
            case 'p': options.PRUNENONSUBSUMED = std::atoi(optarg); break;
            case 'j': constraintfile = optarg; break;
            case 'I': g_inplace = true; break;
            case 'L':  // reference :571-574: the data file is a list of one pattern per line; unindexed, and -t 1 (:677-678)
                options.DOPATTERNPERLINE = true;
                unindexed                = true;
                break;
            case 'F':  // reference :550-559: "S" = from skipgrams (implies -s); a number = from co-occurrence (not in this build)
                if (std::string(optarg) != "S") {
                    std::cerr << "ERROR: option -F " << optarg << " (flexgrams from co-occurrence) is not part of the MI355X-accelerated build (see DESIGN.md, out of scope)" << std::endl;
                    return 2;
                }
                g_flexfromskip      = true;
                options.DOSKIPGRAMS = true;
                break;
            case 1001: g_relations = "skipcontent"; break;
            case 1002: g_relations = "instances"; break;
            case 1003: g_relations = "templates"; break;
            case 'h': usage(); return 0;
            default:
                std::cerr << "ERROR: option -" << (char)(c == '?' ? optopt : c) << " selects a path that is not part of the MI355X-accelerated build (see DESIGN.md, out of scope)" << std::endl;
                return 2;
        }
    }
    if (corpusfile.empty() && inputmodel.empty()) {
        usage();
        return 2;
    }
    if (options.DOPATTERNPERLINE) options.MINTOKENS = 1;
    try {
        ClassDecoder* decoder = NULL;
        ClassDecoder  loaded;
        if (!classfile.empty()) {
            loaded.load(classfile);
            decoder = &loaded;
        }
        if (!g_relations.empty() && decoder == NULL) {  // the reference needs the class encoder here (src/patternmodeller.cpp:275-277)
            std::cerr << "ERROR: --" << g_relations << " needs a class file (--classfile)" << std::endl;
            return 2;
        }
        if (g_flexfromskip && !inputmodel.empty()) {
            std::cerr << "ERROR: -F S on a loaded model needs the reference's trainskipgrams on that model, which is not part of this build; build the model from the corpus (-f) with -F S" << std::endl;
            return 2;
        }
        if (g_inplace && (inputmodel.empty() || corpusfile.empty())) {
            std::cerr << "ERROR: Corpus data file (--datafile|-f) and input model (--inputmodel|-i) must be specified when --constrained|-I is set!." << std::endl;
            return 2;
        }
        if (!constraintfile.empty()) {  // reference :712-718: the constraint is loaded as a pattern set under the run's own thresholds
            std::cerr << "Loading constraint model (aka training/intersection model)" << std::endl;
            g_constraint = new PatternSetModel(constraintfile, options);
            std::cerr << " (Contains " << g_constraint->size() << " patterns)" << std::endl;
        }
        if (unindexed) {
            if (options.DOSKIPGRAMS) {  // unindexed models can only do this exhaustively, on a loaded corpus (reference src/patternmodeller.cpp:723-737)
                options.DOSKIPGRAMS            = false;
                options.DOSKIPGRAMS_EXHAUSTIVE = true;
            }
            if (inputmodel.empty() && options.DOSKIPGRAMS_EXHAUSTIVE) {
                IndexedCorpus          corpus(corpusfile);
                PatternModel<uint32_t> model(&corpus);
                return run(model, corpusfile, inputmodel, outputmodel, options, firstsentence, doprint, doreport, nocoverage, dohistogram, decoder);
            }
            PatternModel<uint32_t> model;
            return run(model, corpusfile, inputmodel, outputmodel, options, firstsentence, doprint, doreport, nocoverage, dohistogram, decoder);
        }
        if (twostage && inputmodel.empty()) {
            // The reference builds an unindexed model first (<out>.stage1) and then an indexed one constrained by it, to bound its memory
            // (src/patternmodeller.cpp:627-663, :756-831). Here the indexed model is built directly — the device holds the forward index
            // anyway — and the observable result of the reference's second stage is reproduced: the same patterns and references, no
            // skipgrams even with -s, and a type count equal to the number of patterns.
            if (outputmodel.empty()) {
                std::cerr << "ERROR: An output model file (--outputmodel) is mandatory for two-stage building!" << std::endl;
                return 2;
            }
            std::cerr << "********* STARTING STAGE 1/2: Building intermediary unindexed patternmodel ******" << std::endl;
            PatternModelOptions stage1 = options;
            stage1.DOSKIPGRAMS = stage1.DOSKIPGRAMS_EXHAUSTIVE = false;
            {
                PatternModel<uint32_t> m1;
                m1.train(corpusfile, stage1, NULL, NULL, false, firstsentence);
                m1.write(outputmodel + ".stage1");
            }
            std::cerr << "********* STARTING STAGE 2/2: Building indexed patternmodel ******" << std::endl;
            IndexedCorpus         corpus(corpusfile);
            IndexedPatternModel<> model(&corpus);
            model.train(corpusfile, stage1, NULL, NULL, false, firstsentence);
            model.settypes_inplace_rebuild();
            std::cerr << "Writing model to " << outputmodel << std::endl;
            model.write(outputmodel);
            std::cerr << "Generating desired views..." << std::endl;
            if (doprint && decoder != NULL) model.print(std::cout, *decoder);
            if (doreport) model.report(std::cout, nocoverage);
            if (dohistogram) model.histogram(std::cout);
            return 0;
        }
        if (inputmodel.empty()) {
            IndexedCorpus         corpus(corpusfile);  // indexed models are built on a loaded corpus (reference :735-737)
            IndexedPatternModel<> model(&corpus);
            return run(model, corpusfile, inputmodel, outputmodel, options, firstsentence, doprint, doreport, nocoverage, dohistogram, decoder);
        }
        IndexedPatternModel<> model;
        return run(model, corpusfile, inputmodel, outputmodel, options, firstsentence, doprint, doreport, nocoverage, dohistogram, decoder);
    } catch (const std::exception& e) {
        std::cerr << "colibri-patternmodeller: " << e.what() << std::endl;
        return 1;
    }
}
