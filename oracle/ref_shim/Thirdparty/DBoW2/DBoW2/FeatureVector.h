// oracle/ref_shim/.../FeatureVector.h -- TEST INFRASTRUCTURE: DBoW2::FeatureVector is a std::map<NodeId, std::vector<unsigned int>>
// (Thirdparty/DBoW2/DBoW2/FeatureVector.h:24-26); this shadows the real header (found first on the include path) so that the DBoW2
// library itself is not needed to compile the reference's src/ORBmatcher.cc.
#ifdef YGZ_REAL_DBOW2   // boundary build with the reference's real DBoW2: hand over to its own header
#include_next "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#else
#ifndef YGZ_ORACLE_REF_SHIM_FEATUREVECTOR_H
#define YGZ_ORACLE_REF_SHIM_FEATUREVECTOR_H
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
}  // namespace DBoW2
#endif
#endif
