// ORBVocabularyDevice.h -- ygz::DeviceORBVocabulary: the reference's ORBVocabulary (include/ORBVocabulary.h =
// DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) with its batch transform on the device (product code, host side; reference tree only:
// it derives from the reference's own DBoW2 class).
//
// Frame::ComputeBoW (src/Frame.cc:495-500) calls mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4), a VIRTUAL member
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:147-148): constructing the vocabulary as ygz::DeviceORBVocabulary (src/System.cc's
// `new ORBVocabulary()`, one line) sends every frame's descriptors down the tree on the GPU -- Frame.cc, KeyFrame.cc and the loaders are
// untouched.  The tree (parent ids + centroids) is uploaded on first use from the nodes the reference's own loader filled; WordId, idf
// weights, the BowVector / FeatureVector maps, tf-idf accumulation in feature order and the L1 / L2 normalisation stay host-side and follow
// transform()'s own code path (:1151-1238), so results are bit-identical to the CPU class.
#ifndef YGZF_HOST_ORBVOCABULARY_DEVICE_H
#define YGZF_HOST_ORBVOCABULARY_DEVICE_H
#include <mutex>
#include <vector>

#include "ORBVocabulary.h"   // the reference's

struct ygzf_ctx;

namespace ygz {

class DeviceORBVocabulary : public ORBVocabulary {
public:
    DeviceORBVocabulary() {}
    ~DeviceORBVocabulary();
    using ORBVocabulary::transform;
    // Transform a set of descriptors into a bow vector and a feature vector (same contract as the base class)
    void transform(const std::vector<DBoW2::FORB::TDescriptor> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const override;
    // Drops the device copy.  Not required after re-loading / re-training: every transform() compares a fingerprint of the tree (shape +
    // sampled centroids and weights) with the one it uploaded.
    void invalidateDevice();

private:
    bool ensureDevice() const;
    unsigned long long treePrint() const;
    mutable std::mutex mMutex;
    mutable ygzf_ctx *mCtx = nullptr;
    mutable size_t mUploadedNodes = 0;
    mutable unsigned long long mUploadedPrint = 0;
};

}  // namespace ygz
#endif
