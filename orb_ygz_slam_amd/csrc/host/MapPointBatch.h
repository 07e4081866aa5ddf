// MapPointBatch.h -- batch front end of ygz::MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:211-271) -- product code, host side; optional.
// LocalMapping calls the member once per new / fused / culled-around MapPoint in three loops (src/LocalMapping.cc:907, :1205, :1314):
//     for (MapPoint *pMP : points) { pMP->ComputeDistinctiveDescriptors(); ... }
// becomes
//     ygz::ComputeDistinctiveDescriptorsBatch(points);          // ONE device call for all of them (ygzf_distinctive_descriptors_batch)
//     for (MapPoint *pMP : points) { /* the rest of the loop body, e.g. pMP->UpdateNormalAndDepth() */ }
// MapPointBatch.cc also holds the strong definition of the member itself (the link drops the reference's body: objcopy --weaken on MapPoint.o):
// called alone it runs one point through the same device kernel; called from the batch front end it picks up the batch's answer.
#ifndef YGZF_HOST_MAPPOINT_BATCH_H
#define YGZF_HOST_MAPPOINT_BATCH_H
#include <vector>
namespace ygz {
class MapPoint;
void ComputeDistinctiveDescriptorsBatch(const std::vector<MapPoint *> &points);
}
#endif
