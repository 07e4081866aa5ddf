// oracle/ref_shim/.../BowVector.h -- TEST INFRASTRUCTURE: DBoW2::BowVector is a std::map<WordId, WordValue> (Thirdparty/DBoW2/DBoW2/BowVector.h)
#ifdef YGZ_REAL_DBOW2   // boundary build with the reference's real DBoW2: hand over to its own header
#include_next "Thirdparty/DBoW2/DBoW2/BowVector.h"
#else
#ifndef YGZ_ORACLE_REF_SHIM_BOWVECTOR_H
#define YGZ_ORACLE_REF_SHIM_BOWVECTOR_H
#include <map>
namespace DBoW2 {
typedef unsigned int WordId;
typedef double WordValue;
class BowVector : public std::map<WordId, WordValue> {};
}  // namespace DBoW2
#endif
#endif
