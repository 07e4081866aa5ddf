// ORBVocabularyDevice.cc -- see ORBVocabularyDevice.h (product code, host side).
#include "ORBextractor.h"          // first: the replacement header (same include guard as the reference's)
#include "ORBVocabularyDevice.h"

#include <cstdio>
#include <cstring>

#include "../../../include/ygzf.h"
#include "ygzf_pool.h"

namespace ygz {

DeviceORBVocabulary::~DeviceORBVocabulary() { ygzf_destroy(mCtx); }

void DeviceORBVocabulary::invalidateDevice() {
    std::lock_guard<std::mutex> lk(mMutex);
    mUploadedNodes = 0;
    mUploadedPrint = 0;
}

// What identifies the tree the device holds: node count, depth, branching, and the centroid + weight of a spread of nodes (first, last and 62
// in between).  The loaders (loadFromTextFile / loadFromBinaryFile / create) are untouched reference code and cannot tell this class that
// they ran; a reloaded or retrained vocabulary of the same shape differs in these samples.
unsigned long long DeviceORBVocabulary::treePrint() const {
    unsigned long long hsh = 1469598103934665603ull ^ ((unsigned long long) m_nodes.size() << 16) ^ ((unsigned long long) m_L << 8) ^ (unsigned long long) m_k;
    const size_t n = m_nodes.size();
    for (size_t j = 0; j < 64 && n > 0; j++) {
        const size_t i = n <= 64 ? (j < n ? j : n - 1) : j * (n - 1) / 63;
        unsigned long long v[5] = {0, 0, 0, 0, 0};
        if (m_nodes[i].descriptor.cols == 32 && m_nodes[i].descriptor.data) std::memcpy(v, m_nodes[i].descriptor.data, 32);
        std::memcpy(&v[4], &m_nodes[i].weight, sizeof(double) < 8 ? sizeof(double) : 8);
        for (unsigned long long x : v) hsh = (hsh ^ x) * 1099511628211ull;
        hsh = (hsh ^ (unsigned long long) m_nodes[i].parent) * 1099511628211ull;
    }
    return hsh ? hsh : 1;
}

bool DeviceORBVocabulary::ensureDevice() const {
    const unsigned long long print = treePrint();
    if (mCtx && mUploadedNodes == m_nodes.size() && mUploadedPrint == print) return true;
    if (!mCtx) {
        ygzf_extractor_cfg cfg = {1000, 1.2f, 8, 20, 7, 0};   // only the context's stream and buffers are used
        if (ygzf_create(ORBextractor::sDevice, &cfg, 64, 64, 1, &mCtx) != YGZF_OK) {
            ygzf_host::report_failure("ygz::DeviceORBVocabulary", ygzf_last_error(nullptr));
            mCtx = nullptr;
            return false;
        }
    }
    const int n = (int) m_nodes.size();
    std::vector<int> parent(n);
    std::vector<uint8_t> desc((size_t) n * 32, 0);
    for (int i = 0; i < n; i++) {
        parent[i] = i == 0 ? -1 : (int) m_nodes[i].parent;
        if (i > 0 && m_nodes[i].descriptor.cols == 32) std::memcpy(&desc[(size_t) i * 32], m_nodes[i].descriptor.data, 32);
    }
    // children order on the device = ascending node id; the loaders push children in that order too (loadFromTextFile :1409-1416, load
    // :1692-1705 for vocabularies saved by DBoW2).  Verified here, node by node: a vocabulary that breaks it is refused with a message that
    // says so (transform() then returns empty vectors: libygzf has no CPU fallback; ORBVocabulary::transform is the host path to call instead).
    for (int i = 0; i < n; i++) {
        const std::vector<DBoW2::NodeId> &ch = m_nodes[i].children;
        for (size_t k = 1; k < ch.size(); k++)
            if (ch[k] < ch[k - 1]) {
                char msg[96];
                snprintf(msg, sizeof msg, "children of node %d are not in ascending id order", i);
                ygzf_host::report_failure("ygz::DeviceORBVocabulary", msg);
                return false;
            }
    }
    if (ygzf_vocabulary_set(mCtx, n, m_L, parent.data(), desc.data()) != YGZF_OK) {
        ygzf_host::report_failure("ygz::DeviceORBVocabulary", ygzf_last_error(mCtx));
        return false;
    }
    mUploadedNodes = m_nodes.size();
    mUploadedPrint = print;
    return true;
}

void DeviceORBVocabulary::transform(const std::vector<DBoW2::FORB::TDescriptor> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const {
    v.clear();
    fv.clear();
    if (empty()) return;
    const int n = (int) features.size();
    std::vector<int> leaf(n), nid(n);
    {
        std::lock_guard<std::mutex> lk(mMutex);   // Tracking and LoopClosing threads share the vocabulary object
        std::vector<uint8_t> desc((size_t) n * 32);
        for (int i = 0; i < n; i++) std::memcpy(&desc[(size_t) i * 32], features[i].data, 32);
        if (!ensureDevice()) {   // the reason (no device, upload failure, children not in id order) is already on stderr
            ygzf_host::report_failure("ygz::DeviceORBVocabulary::transform", "the vocabulary is not on the device: BowVector / FeatureVector stay EMPTY for this frame "
                                                                                "(no CPU fallback; ORBVocabulary::transform is the host path)");
            return;
        }
        if (ygzf_bow_transform(mCtx, n, desc.data(), levelsup, leaf.data(), nid.data()) != YGZF_OK) {
            ygzf_host::report_failure("ygz::DeviceORBVocabulary::transform (BowVector / FeatureVector stay EMPTY for this frame)", ygzf_last_error(mCtx));
            return;
        }
    }
    // from here: transform(features, v, fv, levelsup) of the base class with the per-feature descent replaced by the device result
    DBoW2::LNorm norm;
    const bool must = m_scoring_object->mustNormalize(norm);
    if (m_weighting == DBoW2::TF || m_weighting == DBoW2::TF_IDF) {
        for (int i = 0; i < n; i++) {
            const DBoW2::WordId id = m_nodes[leaf[i]].word_id;
            const DBoW2::WordValue w = m_nodes[leaf[i]].weight;
            if (w > 0) {   // not stopped
                v.addWeight(id, w);
                fv.addFeature((DBoW2::NodeId) nid[i], (unsigned) i);
            }
        }
        if (!v.empty() && !must) {
            const double nd = v.size();
            for (DBoW2::BowVector::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
        }
    } else {   // IDF || BINARY
        for (int i = 0; i < n; i++) {
            const DBoW2::WordId id = m_nodes[leaf[i]].word_id;
            const DBoW2::WordValue w = m_nodes[leaf[i]].weight;
            if (w > 0) {
                v.addIfNotExist(id, w);
                fv.addFeature((DBoW2::NodeId) nid[i], (unsigned) i);
            }
        }
    }
    if (must) v.normalize(norm);
}

}  // namespace ygz
