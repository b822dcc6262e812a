// colibri-classencode (MI355X build) — the reference's class-encoder driver on the accelerated path: same flags, file names and
// messages as reference src/classencode.cpp:47-198 for  -c -o -d -l -u -e -U -t -f ; the flags that select paths outside the
// accelerated subset (-n -F -V, bz2 / FoLiA input) are reported and rejected instead of being silently ignored.
#include <getopt.h>

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "classencoder.h"

namespace {
void usage() {
    std::cerr << "colibri-classencode (MI355X-native build of the Colibri Core class encoder)\n"
                 "Syntax: colibri-classencode [ -c classmodel ] corpus [corpus2 etc..]\n"
                 "Description: Encodes a corpus. If used with -c, encodes a corpus according to the specified pre-existing class model\n"
                 "The corpus file should be plain text, preferably tokenised (tokens space delimited), one sentence per line, unix newlines.\n"
                 "Options: -o    outputprefix for class file\n"
                 "         -d    output directory, including trailing slash\n"
                 "         -l    read input filenames from list-file (one filename per line)\n"
                 "         -u    produce one unified encoded corpus (in case multiple corpora are specified)\n"
                 "         -e    extend specified class file with unseen classes\n"
                 "         -U    encode all unseen classes using one special unknown class\n"
                 "         -t    word occurrence threshold (default: 1)\n";
}
void strip_extension(std::string& filename, const std::string& extension) {  // reference src/common.cpp: strips ".<extension>" when it ends the name
    const std::string suffix = "." + extension;
    if (filename.size() > suffix.size() && filename.compare(filename.size() - suffix.size(), suffix.size(), suffix) == 0) filename.resize(filename.size() - suffix.size());
}
}  // namespace

int main(int argc, char* argv[]) {
    std::string              classfile, corpusfile, outputprefix, outputdirectoryprefix;
    std::vector<std::string> corpusfiles;
    bool                     unified = false, extend = false, allowunknown = false;
    unsigned int             threshold = 0;
    int                      c;
    while ((c = getopt(argc, argv, "f:hc:o:d:ul:eUt:F:V:n")) != -1) {
        switch (c) {
            case 'f': corpusfiles.push_back(optarg); break;  // kept for backward compatibility, as in the reference
            case 'c': classfile = optarg; break;
            case 'o': outputprefix = optarg; break;
            case 'd': outputdirectoryprefix = optarg; break;
            case 'u': unified = true; break;
            case 'e': extend = true; break;
            case 'U': allowunknown = true; break;
            case 't': threshold = (unsigned int)std::atoi(optarg); break;
            case 'l': {
                std::ifstream listfile(optarg);
                std::string   line;
                while (std::getline(listfile, line))
                    if (!line.empty()) corpusfiles.push_back(line);
                break;
            }
            case 'h': usage(); return 0;
            case 'n':
            case 'F':
            case 'V':
                std::cerr << "ERROR: option -" << (char)c << " selects a path that is not part of the MI355X-accelerated build (see DESIGN.md, out of scope)" << std::endl;
                return 2;
            default: std::cerr << "Unknown option: -" << (char)optopt << std::endl; return 2;
        }
    }
    for (int i = optind; i < argc; i++) corpusfiles.push_back(argv[i]);
    if (corpusfiles.empty()) {
        usage();
        return 2;
    }
    corpusfile = corpusfiles[0];
    if (outputprefix.empty()) {
        outputprefix = corpusfile.find_last_of("/") == std::string::npos ? corpusfile : corpusfile.substr(corpusfile.find_last_of("/") + 1);
        strip_extension(outputprefix, "bz2");
        strip_extension(outputprefix, "xml");
        strip_extension(outputprefix, "txt");
    }
    try {
        ClassEncoder      classencoder;
        const std::string prefixed = outputdirectoryprefix + outputprefix;
        if (!classfile.empty()) {
            std::cerr << "Loading classes from file" << std::endl;
            classencoder = ClassEncoder(classfile);
            if (extend) {
                std::cerr << "Building classes from corpus (extending existing classes)" << std::endl;
                classencoder.build(corpusfiles, false, threshold, "");
                classencoder.save(prefixed + ".colibri.cls");
                std::cerr << "Built " << prefixed << ".colibri.cls , extending " << classfile << std::endl;
            }
        } else {
            std::cerr << "Building classes from corpus" << std::endl;
            classencoder.build(corpusfiles, false, threshold, "");
            classencoder.save(prefixed + ".colibri.cls");
            std::cerr << "Built " << prefixed << ".colibri.cls" << std::endl;
        }
        const unsigned int highestclass = classencoder.gethighestclass();
        for (size_t i = 0; i < corpusfiles.size(); i++) {
            std::string outfile = corpusfiles[i];
            if (outfile.find_last_of("/") != std::string::npos) outfile = outfile.substr(outfile.find_last_of("/") + 1);
            if (unified) outfile = outputprefix;
            strip_extension(outfile, "bz2");
            strip_extension(outfile, "txt");
            strip_extension(outfile, "xml");
            if (!outputdirectoryprefix.empty()) outfile = outputdirectoryprefix + "/" + outfile;
            std::cerr << "Encoding corpus " << corpusfiles[i] << " to " << outfile << ".colibri.dat" << std::endl;
            classencoder.encodefile(corpusfiles[i], outfile + ".colibri.dat", allowunknown, extend, unified && i > 0, false);
            std::cerr << "...Done" << std::endl;
        }
        if (classencoder.gethighestclass() > highestclass) {
            if (extend) {
                classencoder.save(outputprefix + ".colibri.cls");
                std::cerr << "Built " << outputprefix << ".colibri.cls" << std::endl;
            } else {
                std::cerr << "WARNING: classes were added but the result was ignored! Use -e!" << std::endl;
            }
        }
    } catch (const UnknownTokenError&) {
        std::cerr << "ERROR: the corpus contains a word that has no class; pass -U (one unknown class) or -e (extend the classes)" << std::endl;
        return 4;
    } catch (const std::exception& e) {
        std::cerr << "colibri-classencode: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
