// classdecoder.h — class id -> word map, only what printing a model needs.
// reference: include/classdecoder.h (ClassDecoder), src/classdecoder.cpp:20-43 (bytestoint), :84-130 (load), :259-284 (getdataversion).
// The .colibri.cls format is one "<class id>\t<word>" per line.
#ifndef COLIBRI_AMD_CLASSDECODER_H
#define COLIBRI_AMD_CLASSDECODER_H
#include <istream>
#include <string>
#include <unordered_map>

/** decodes one little-endian base-128 class id (high bit set on all bytes but the last); integer shifts, no pow() */
unsigned int bytestoint(const unsigned char* a, unsigned int* length = NULL);
/** 2 for files starting with A2 <version>; 1 for header-less v1 data (first byte is a token length) */
unsigned char getdataversion(std::istream& in);

class ClassDecoder {
  public:
    static const unsigned char delimiterclass = 0, boundaryclass = 1, unknownclass = 2, skipclass = 3, flexclass = 4;
    ClassDecoder();
    explicit ClassDecoder(const std::string& filename);
    void load(const std::string& filename);
    bool hasclass(unsigned int cls) const { return classes.count(cls) != 0; }
    const std::string& operator[](unsigned int cls) const;
    size_t size() const { return classes.size(); }
    unsigned int gethighestclass() const { return highestclass; }

  private:
    std::unordered_map<unsigned int, std::string> classes;
    unsigned int highestclass;
};
#endif
