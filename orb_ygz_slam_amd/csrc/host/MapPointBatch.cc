// MapPointBatch.cc -- see MapPointBatch.h (product code, host side).  Compiled inside the reference tree against the reference's own,
// unchanged include/MapPoint.h / KeyFrame.h.
#include "ORBextractor.h"   // first: inside the reference tree this is the replacement header (same include guard)
#include "MapPointBatch.h"
#include "ygz_compat.h"

#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../../../include/ygzf.h"
#include "ygzf_pool.h"

namespace {
// what the batch front end worked out for a point: the observation rows it saw (the member checks that it sees the same ones) and the winner
struct Answer {
    std::vector<std::pair<ygz::KeyFrame *, size_t> > rows;
    int best = -1;
};
thread_local std::map<ygz::MapPoint *, Answer> *t_answers = nullptr;

// the rows the reference's member collects (:224-232): observations in map order, KeyFrames that are not bad
void rows_of(const std::map<ygz::KeyFrame *, size_t> &observations, std::vector<std::pair<ygz::KeyFrame *, size_t> > &rows) {
    rows.clear();
    rows.reserve(observations.size());
    for (std::map<ygz::KeyFrame *, size_t>::const_iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++)
        if (!mit->first->isBad()) rows.push_back(std::make_pair(mit->first, mit->second));
}

bool device_best(int nPoints, const std::vector<int> &off, const std::vector<uint8_t> &desc, std::vector<int> &best, const char *who) {
    best.assign(nPoints, -1);
    ygzf_host::Lease lease(ygz::ORBextractor::sDevice);
    if (!lease) return false;
    if (ygzf_distinctive_descriptors_batch(lease.get(), nPoints, off.data(), desc.data(), best.data()) != YGZF_OK) {
        ygzf_host::report_failure(who, ygzf_last_error(lease.get()));
        return false;
    }
    return true;
}
}  // namespace

namespace ygz {

// src/MapPoint.cc:211-271: same snapshot, same row order, same result; the all-pairs distances and the per-row medians run on the device.
void MapPoint::ComputeDistinctiveDescriptors() {
    std::map<KeyFrame *, size_t> observations;
    {
        std::unique_lock<std::mutex> lock1(mMutexFeatures);
        if (mbBad) return;
        observations = mObservations;
    }
    if (observations.empty()) return;
    std::vector<std::pair<KeyFrame *, size_t> > rows;
    rows_of(observations, rows);
    if (rows.empty()) return;
    int best = -1;
    if (t_answers) {
        std::map<MapPoint *, Answer>::const_iterator it = t_answers->find(this);
        if (it != t_answers->end() && it->second.rows == rows) best = it->second.best;     // nothing changed since the batch gathered this point
    }
    if (best < 0) {
        std::vector<int> off(2, 0), b;
        off[1] = (int) rows.size();
        std::vector<uint8_t> desc(rows.size() * 32);
        for (size_t i = 0; i < rows.size(); i++) std::memcpy(&desc[i * 32], rows[i].first->mDescriptors.ptr(rows[i].second), 32);
        if (!device_best(1, off, desc, b, "ygz::MapPoint::ComputeDistinctiveDescriptors")) return;   // the descriptor stays what it was
        best = b[0];
    }
    if (best < 0 || best >= (int) rows.size()) return;
    {
        std::unique_lock<std::mutex> lock(mMutexFeatures);
        mDescriptor = rows[best].first->mDescriptors.row(rows[best].second).clone();
    }
}

void ComputeDistinctiveDescriptorsBatch(const std::vector<MapPoint *> &points) {
    std::map<MapPoint *, Answer> answers;
    std::vector<MapPoint *> order;
    std::vector<int> off(1, 0);
    std::vector<uint8_t> desc;
    for (MapPoint *mp : points) {
        if (!mp || mp->isBad() || answers.count(mp)) continue;
        Answer a;
        rows_of(mp->GetObservations(), a.rows);
        if (a.rows.empty()) continue;
        const size_t d0 = desc.size();
        desc.resize(d0 + a.rows.size() * 32);
        for (size_t i = 0; i < a.rows.size(); i++) std::memcpy(&desc[d0 + i * 32], a.rows[i].first->mDescriptors.ptr(a.rows[i].second), 32);
        off.push_back((int) (desc.size() / 32));
        order.push_back(mp);
        answers[mp] = a;
    }
    if (!order.empty()) {
        std::vector<int> best;
        if (device_best((int) order.size(), off, desc, best, "ygz::ComputeDistinctiveDescriptorsBatch"))
            for (size_t p = 0; p < order.size(); p++) answers[order[p]].best = best[p];
    }
    // the member applies the answers under the MapPoint's own lock (a point whose observations changed meanwhile takes the one-point path)
    t_answers = &answers;
    for (MapPoint *mp : points)
        if (mp) mp->ComputeDistinctiveDescriptors();
    t_answers = nullptr;
}

}  // namespace ygz
