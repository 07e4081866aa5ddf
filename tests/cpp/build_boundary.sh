#!/bin/bash
# tests/cpp/build_boundary.sh -- builds tests/cpp/bin/boundary_frame: the REFERENCE's own src/Frame.cc, compiled where it lies with the real
# include/Frame.h, on top of the PRODUCT's class shells (orb_ygz_slam_amd/csrc/host) and libygzf.so.  This is the drop-in recipe of
# INTEGRATION.md executed for real:
#   * orb_ygz_slam_amd/csrc/host/ORBextractor.h takes the place of the reference's include/ORBextractor.h (same include guard; here it
#     is force-included ahead of everything, in a checkout one copies it over the file), ORBextractor.cc that of src/ORBextractor.cc;
#   * ORBmatcher.cc / SparseImageAlign.cc are compiled against the reference's OWN, unchanged include/ORBmatcher.h, SparseImageAlign.h,
#     NLSSolver.h and define those classes' hot-path members;
#   * the vocabulary is the reference's REAL DBoW2 (Thirdparty/DBoW2, compiled in), wrapped by ygz::DeviceORBVocabulary, whose virtual
#     transform() Frame::ComputeBoW reaches unchanged;
#   * the reference's own src/Tracking.cc is compiled where it lies, unchanged, against the product's ORBextractor.h and the reference's own
#     ORBmatcher.h / SparseImageAlign.h / Frame.h, and linked in: its calls of ORBextractor / ORBmatcher::SearchBy* / FindDirectProjection /
#     SparseImgAlign::run bind to the product's strong definitions.  Everything outside the hot path that it names (System, LocalMapping,
#     LoopClosing, Optimizer, Initializer, PnPsolver, Map, KeyFrame machinery, viewer) is declarations (oracle/ref_shim/tracking/) whose
#     members the link resolves to ONE aborting stand-in: the alias list is generated from the link's own undefined-symbol report;
#   * OpenCV, Eigen and Sophus are not installed in this environment: oracle/ref_shim stands in for their headers (test infrastructure);
#     its compute primitives are replaced by aborting bodies (tests/cpp/mini_cv_nocompute.cpp), the pose algebra behind the Sophus stand-in
#     is oracle_align.cpp's.
# Needs the reference checkout (REF, default /root/reference); the binary travels to the GPU box (tests/cpp/bin is git-ignored only).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
REF=${REF:-/root/reference}
H=$ROOT/orb_ygz_slam_amd/csrc/host
S=$ROOT/oracle/ref_shim
OUT=$ROOT/tests/cpp/bin
if [ ! -f "$REF/src/Frame.cc" ]; then echo "reference checkout absent: keeping prebuilt $OUT/boundary_frame (if any)"; exit 0; fi
mkdir -p "$OUT"
FLAGS="-O2 -std=c++14 -msse4.2 -pthread -w -ffp-contract=off -DYGZ_REF_TRACKING -DYGZ_REF_MATCHER -DYGZ_REF_FRAME -DYGZ_BOUNDARY_BUILD -DYGZ_REAL_DBOW2 -DYGZF_WITH_REFERENCE_HEADERS"
INC="-I$S/tracking -I$S -I$ROOT/oracle -I$REF/include -I$REF -I$H -include $S/dbow2_stubs.h -include $H/ORBextractor.h -include $S/tracking/tracking_stubs.h"
# The reference's own src/ORBmatcher.cc (+ src/Align.cc) stays in the link for the members outside the hot path (Fuse x2, SearchBySim3,
# SearchByProjection(KF, Scw, ...), SearchByBoW(KF, KF, ...)): its definitions are made WEAK, so the strong
# definitions of the hot-path members in the product's ORBmatcher.cc win at link time and nothing in the reference file is edited.
g++ $FLAGS $INC -c "$REF/src/ORBmatcher.cc" -o "$OUT/ref_ORBmatcher.o"
g++ $FLAGS $INC -c "$REF/src/Align.cc" -o "$OUT/ref_Align.o"
# (the reference's OWN body of SearchForTriangulation stays callable as ygz_ref_ORBmatcher_SearchForTriangulation through a renamed, all-weak copy
# of the object: the driver runs it beside the product's strong definition, as with the three batch bindings below)
TRI=_ZN3ygz10ORBmatcher22SearchForTriangulationEPNS_8KeyFrameES2_RN5Eigen8Matrix3fERSt6vectorISt4pairImmESaIS8_EEb
objcopy --redefine-sym $TRI=ygz_ref_ORBmatcher_SearchForTriangulation --weaken "$OUT/ref_ORBmatcher.o" "$OUT/ref_ORBmatcher_orig.o"
objcopy --weaken "$OUT/ref_ORBmatcher.o"
g++ $FLAGS $INC -c "$REF/src/Tracking.cc" -o "$OUT/ref_Tracking.o"
g++ $FLAGS $INC -c "$REF/src/Frame.cc" -o "$OUT/ref_Frame.o"
# The BATCH bindings (TrackingBatched.cc, FrameStereo.cc) are strong definitions of Tracking::SearchLocalPoints, Tracking::SearchLocalPointsDirect and
# Frame::ComputeStereoMatches: the reference's objects are weakened, so the product's bodies win and every other member stays the reference's.
# The reference's OWN bodies of those three stay callable for the comparison the driver makes (same objects, same inputs, per-call form against
# batch form): a second, all-weak copy of each object in which just these symbols carry another name (its duplicates of everything else lose
# against the strong originals).
objcopy --redefine-sym _ZN3ygz8Tracking17SearchLocalPointsEv=ygz_ref_Tracking_SearchLocalPoints \
        --redefine-sym _ZN3ygz8Tracking23SearchLocalPointsDirectEv=ygz_ref_Tracking_SearchLocalPointsDirect --weaken "$OUT/ref_Tracking.o" "$OUT/ref_Tracking_orig.o"
objcopy --redefine-sym _ZN3ygz5Frame20ComputeStereoMatchesEv=ygz_ref_Frame_ComputeStereoMatches --weaken "$OUT/ref_Frame.o" "$OUT/ref_Frame_orig.o"
objcopy --weaken-symbol=_ZN3ygz8Tracking17SearchLocalPointsEv --weaken-symbol=_ZN3ygz8Tracking23SearchLocalPointsDirectEv "$OUT/ref_Tracking.o"
objcopy --weaken-symbol=_ZN3ygz5Frame20ComputeStereoMatchesEv "$OUT/ref_Frame.o"
OBJS="$OUT/ref_Tracking.o $OUT/ref_Tracking_orig.o $OUT/ref_Frame.o $OUT/ref_Frame_orig.o $OUT/ref_ORBmatcher.o $OUT/ref_ORBmatcher_orig.o $OUT/ref_Align.o"
SRCS="$H/TrackingBatched.cc $H/FrameStereo.cc $H/ORBextractor.cc $H/ORBmatcher.cc $H/SparseImageAlign.cc $H/ORBVocabularyDevice.cc $H/ygzf_pool.cc \
    $REF/Thirdparty/DBoW2/DBoW2/FORB.cpp $REF/Thirdparty/DBoW2/DBoW2/BowVector.cpp $REF/Thirdparty/DBoW2/DBoW2/FeatureVector.cpp \
    $REF/Thirdparty/DBoW2/DBoW2/ScoringObject.cpp $REF/Thirdparty/DBoW2/DUtils/Random.cpp $REF/Thirdparty/DBoW2/DUtils/Timestamp.cpp \
    $ROOT/tests/cpp/boundary_frame.cc $ROOT/tests/cpp/mini_cv_nocompute.cpp $ROOT/oracle/oracle_align.cpp $ROOT/oracle/oracle_direct.cpp"
i=0
for f in $SRCS; do i=$((i+1)); g++ $FLAGS $INC -c "$f" -o "$OUT/b_$i.o"; OBJS="$OBJS $OUT/b_$i.o"; done
LINK="-L$ROOT/orb_ygz_slam_amd/lib -lygzf -Wl,-rpath,\$ORIGIN/../../../orb_ygz_slam_amd/lib"
# what the link still misses = the members of the classes outside the hot path: each becomes an alias of one aborting function
cat > "$OUT/outside.S" <<'ASM'
    .text
    .globl ygz_outside_the_hot_path
    .type ygz_outside_the_hot_path, @function
ygz_outside_the_hot_path:
    jmp ygz_outside_abort@PLT
ASM
cat > "$OUT/outside_abort.cc" <<'CC'
#include <cstdio>
#include <cstdlib>
extern "C" void ygz_outside_abort() { std::fprintf(stderr, "boundary build: a member of a class outside the hot path was called (aborting stand-in)\n"); std::abort(); }
CC
g++ -O2 -c "$OUT/outside_abort.cc" -o "$OUT/outside_abort.o"
( g++ -pthread $OBJS "$OUT/outside_abort.o" $LINK -Wl,--no-demangle -o "$OUT/boundary_frame.try" 2>&1 || true ) | grep -o "undefined reference to \`[^']*'" | sed "s/undefined reference to \`//; s/'\$//" | sort -u > "$OUT/outside.syms"
while read -r sym; do
    case "$sym" in
        _ZN3ygz*|_ZNK3ygz*) printf '    .globl %s\n    .set %s, ygz_outside_the_hot_path\n' "$sym" "$sym" >> "$OUT/outside.S" ;;
        *) echo "unexpected undefined symbol outside namespace ygz: $sym"; exit 1 ;;
    esac
done < "$OUT/outside.syms"
g++ -c "$OUT/outside.S" -o "$OUT/outside.o"
g++ -pthread $OBJS "$OUT/outside_abort.o" "$OUT/outside.o" $LINK -o "$OUT/boundary_frame"
cp "$OUT/outside.syms" "$OUT/boundary_frame.outside"
rm -f $OBJS "$OUT/outside.S" "$OUT/outside.o" "$OUT/outside_abort.cc" "$OUT/outside_abort.o" "$OUT/outside.syms" "$OUT/boundary_frame.try"
rm -f "$OUT/ref_ORBmatcher.o" "$OUT/ref_ORBmatcher_orig.o" "$OUT/ref_Align.o" "$OUT/ref_Tracking_orig.o" "$OUT/ref_Frame.o" "$OUT/ref_Frame_orig.o"
# strong (T) = the product's definition was linked; weak (W) = the reference's body is still the one in use
nm -C "$OUT/boundary_frame" | grep -E " [TW] ygz::(ORBmatcher::(SearchByProjection|SearchByBoW|SearchForInitialization|FindDirectProjection|Fuse|SearchBySim3|SearchForTriangulation|DescriptorDistance)|SparseImgAlign::run|ORBextractor::operator\(\)|Frame::ComputeStereoMatches|Tracking::(TrackWithSparseAlignment|TrackWithMotionModel|SearchLocalPoints|MonocularInitialization|Relocalization|SearchLocalPointsDirect|TrackReferenceKeyFrame))\(" | sed 's/^[0-9a-f]* //' | sort > "$OUT/boundary_frame.symbols"
echo "built $OUT/boundary_frame"

# ---- second target: tests/cpp/bin/libboundary_mappoint.so -- the reference's own src/MapPoint.cc with the REAL include/MapPoint.h (the recipe of
# oracle/Makefile's ref_mappoint) under the product's MapPointBatch.cc: strong MapPoint::ComputeDistinctiveDescriptors + the batch front end; the
# reference's body stays reachable as ygz_ref_MapPoint_ComputeDistinctiveDescriptors (tests/cpp/boundary_mappoint.cc compares the three forms).
MFLAGS="-O2 -std=c++14 -msse4.2 -fPIC -pthread -w -ffp-contract=off -DYGZ_REF_MATCHER -DYGZ_REF_MAPPOINT -DYGZF_WITH_REFERENCE_HEADERS"
MINC="-I$S -I$ROOT/oracle -I$REF/include -I$REF -I$H -include $S/mini_cv.h"
g++ $MFLAGS $MINC -c "$REF/src/MapPoint.cc" -o "$OUT/m_MapPoint.o"
objcopy --redefine-sym _ZN3ygz8MapPoint29ComputeDistinctiveDescriptorsEv=ygz_ref_MapPoint_ComputeDistinctiveDescriptors --weaken "$OUT/m_MapPoint.o" "$OUT/m_MapPoint_orig.o"
objcopy --weaken-symbol=_ZN3ygz8MapPoint29ComputeDistinctiveDescriptorsEv "$OUT/m_MapPoint.o"
MOBJS="$OUT/m_MapPoint.o $OUT/m_MapPoint_orig.o"
i=0
for f in "$REF/src/ORBmatcher.cc" "$REF/src/Align.cc" "$H/MapPointBatch.cc" "$H/ygzf_pool.cc" "$ROOT/tests/cpp/boundary_mappoint.cc" "$ROOT/oracle/oracle_align.cpp" "$ROOT/oracle/oracle_direct.cpp"; do
    i=$((i+1)); g++ $MFLAGS $MINC -c "$f" -o "$OUT/m_$i.o"; MOBJS="$MOBJS $OUT/m_$i.o"
done
g++ -shared -pthread $MOBJS -L$ROOT/orb_ygz_slam_amd/lib -lygzf -Wl,-rpath,\$ORIGIN/../../../orb_ygz_slam_amd/lib -o "$OUT/libboundary_mappoint.so"
nm -C "$OUT/libboundary_mappoint.so" | grep -E " [TW] (ygz::MapPoint::ComputeDistinctiveDescriptors|ygz::ComputeDistinctiveDescriptorsBatch|ygz_ref_MapPoint)" | sed 's/^[0-9a-f]* //' | sort > "$OUT/libboundary_mappoint.symbols"
rm -f $MOBJS
if ldd -r "$OUT/libboundary_mappoint.so" 2>&1 | grep -q "undefined symbol"; then ldd -r "$OUT/libboundary_mappoint.so" 2>&1 | grep "undefined symbol"; exit 1; fi
echo "built $OUT/libboundary_mappoint.so"
