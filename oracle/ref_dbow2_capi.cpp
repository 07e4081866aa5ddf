// oracle/ref_dbow2_capi.cpp -- TEST INFRASTRUCTURE: the REFERENCE's own DBoW2 (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h, FORB.cpp, BowVector.cpp,
// FeatureVector.cpp, ScoringObject.cpp, compiled where they lie into oracle/_ref/libref_dbow2.so over oracle/ref_shim) behind flat C entry
// points: a vocabulary read by the reference's loadFromTextFile and ORBVocabulary::transform(features, BowVector, FeatureVector, levelsup)
// -- the call Frame::ComputeBoW makes (src/Frame.cc:495-500, levelsup = 4).
#include <opencv2/core/core.hpp>

#include "Thirdparty/DBoW2/DBoW2/FORB.h"
#include "Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h"

typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> ORBVocabulary;   // include/ORBVocabulary.h

extern "C" {

void *yr_voc_load_text(const char *path) {
    ORBVocabulary *v = new ORBVocabulary();
    if (!v->loadFromTextFile(path)) { delete v; return nullptr; }
    return v;
}
void yr_voc_free(void *v) { delete (ORBVocabulary *) v; }
int yr_voc_size(void *v) { return (int) ((ORBVocabulary *) v)->size(); }

// transform(features, v, fv, levelsup): BowVector as (word id, value) in map order, FeatureVector as (node id, offsets into feat_idx)
int yr_voc_transform(void *voc, const uint8_t *desc, int n, int levelsup, int *bow_ids, double *bow_vals, int bow_cap, int *n_bow, int *fv_nodes, int *fv_off,
                     int *fv_idx, int fv_cap, int *n_fv) {
    std::vector<cv::Mat> feats;
    for (int i = 0; i < n; i++) {
        cv::Mat m(1, 32, CV_8U);
        std::memcpy(m.data, desc + 32 * (size_t) i, 32);
        feats.push_back(m);
    }
    DBoW2::BowVector bv;
    DBoW2::FeatureVector fv;
    ((ORBVocabulary *) voc)->transform(feats, bv, fv, levelsup);
    if ((int) bv.size() > bow_cap || (int) fv.size() > fv_cap) return -1;
    int k = 0;
    for (auto &e : bv) { bow_ids[k] = (int) e.first; bow_vals[k] = e.second; k++; }
    *n_bow = k;
    k = 0;
    int pos = 0;
    for (auto &e : fv) {
        fv_nodes[k] = (int) e.first;
        fv_off[k] = pos;
        for (unsigned f : e.second) fv_idx[pos++] = (int) f;
        k++;
    }
    fv_off[k] = pos;
    *n_fv = k;
    return 0;
}

}  // extern "C"
