// common.h — exceptions and small constants of the C++ face (names as in the reference so that callers compile unchanged).
// reference: include/common.h:27-28 (bitmask), :41-54 (InternalError, KeyError, UnknownTokenError)
#ifndef COLIBRI_AMD_COMMON_H
#define COLIBRI_AMD_COMMON_H
#include <cstdint>
#include <stdexcept>
#include <string>

class InternalError : public std::runtime_error {
  public:
    InternalError() : std::runtime_error("Colibri internal error") {}
    explicit InternalError(const std::string& msg) : std::runtime_error(msg) {}
};
class KeyError : public std::runtime_error {
  public:
    KeyError() : std::runtime_error("Colibri KeyError") {}
};
class UnknownTokenError : public std::runtime_error {
  public:
    UnknownTokenError() : std::runtime_error("The input contained an unknown token") {}
};

// bit i = token i of a pattern is a gap
inline uint32_t bitmask_of(int i) { return i < 31 ? (uint32_t(1) << i) : 0u; }

enum PatternCategory { UNKNOWNPATTERN = 0, NGRAM = 1, SKIPGRAM = 2, FLEXGRAM = 3, SKIPGRAMORFLEXGRAM = 4 };

// reserved classes (reference include/classdecoder.h:48-52)
namespace colibri_classes {
constexpr unsigned char delimiterclass = 0, boundaryclass = 1, unknownclass = 2, skipclass = 3, flexclass = 4;
}
#endif
