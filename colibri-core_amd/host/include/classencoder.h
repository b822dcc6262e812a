// classencoder.h — ClassEncoder of the C++ face: same names, arguments and behaviour as the reference's class for the flows
// colibri-classencode uses (reference include/classencoder.h:51-260, src/classencoder.cpp), with the corpus-proportional work —
// the word frequency list (processcorpus :156-188) and the encoding of the corpus (encodefile / encodestring :369-436, :550-600) —
// done by libcolibri_hip.so (colibri_text_*). What stays on the host is proportional to the vocabulary: classes by frequency
// (buildclasses :213-229), the class file (load :94-132, save :270-277) and the unknown-word policy per DISTINCT word.
// Ties between equally frequent words fall as in the reference because the same libstdc++ containers are filled in the same
// order (first occurrence in the text, which the device reports). No CPU fallback: without a device every corpus call throws.
// Outside the accelerated subset (reported with InternalError): minlength / maxlength, vocabulary files, -n (ignore newlines),
// bz2 / FoLiA input.
#ifndef COLIBRI_AMD_CLASSENCODER_H
#define COLIBRI_AMD_CLASSENCODER_H
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "common.h"

class ClassEncoder {
  private:
    std::unordered_map<std::string, unsigned int> classes;
    unsigned int                                  highestclass;
    unsigned int                                  minlength, maxlength;

  public:
    static const unsigned char delimiterclass = 0;
    static const unsigned char boundaryclass  = 1;
    static const unsigned char unknownclass   = 2;
    static const unsigned char skipclass      = 3;
    static const unsigned char flexclass      = 4;

    ClassEncoder(const unsigned int minlength = 0, const unsigned int maxlength = 0);                               ///< empty encoder, classes start at 6 (reference :83-88)
    ClassEncoder(const std::string& filename, const unsigned int minlength = 0, const unsigned int maxlength = 0);  ///< load a class file (:90-92)
    void load(const std::string& filename, const unsigned int minlength = 0, const unsigned int maxlength = 0);     ///< (:94-132)

    /** word frequencies of one corpus / several corpora on the device, then buildclasses (:231-268) */
    void build(const std::string& filename, unsigned int threshold = 0, const std::string& vocabfile = "");
    void build(const std::vector<std::string>& files, bool quiet = false, unsigned int threshold = 0, const std::string& vocabfile = "");
    /** adds the corpus' word counts to freqlist; new words are inserted in the order of their first occurrence (:134-188) */
    void processcorpus(const std::string& filename, std::unordered_map<std::string, unsigned int>& freqlist, std::unordered_set<std::string>* vocab = NULL);
    /** classes by descending frequency for the words at or above the threshold that have none yet (:213-229) */
    void buildclasses(const std::unordered_map<std::string, unsigned int>& freqlist, unsigned int threshold = 0);
    void save(const std::string& filename);  ///< "class TAB word" per line (:270-277)

    /** plain text -> .colibri.dat (v2) on the device (:466-548, :550-600). Unknown words: new classes when autoaddunknown, the unknown
     *  class when allowunknown, else UnknownTokenError. */
    void encodefile(const std::string& inputfilename, const std::string& outputfilename, bool allowunknown, bool autoaddunknown = false, bool append = false,
                    bool ignorenewlines = false, bool quiet = false);

    int          size() const { return (int)classes.size(); }
    unsigned int gethighestclass() const { return highestclass; }
    unsigned int operator[](const std::string& key) {
        auto it = classes.find(key);
        if (it == classes.end()) throw KeyError();
        return it->second;
    }
    void add(const std::string& s, const unsigned int cls) {
        classes[s] = cls;
        if (cls > highestclass) highestclass = cls;
    }
    typedef std::unordered_map<std::string, unsigned int>::const_iterator const_iterator;
    const_iterator begin() const { return classes.begin(); }
    const_iterator end() const { return classes.end(); }
};
#endif
