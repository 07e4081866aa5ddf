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
FLAGS="-O2 -std=c++14 -msse4.2 -pthread -w -ffp-contract=off -DYGZ_REF_MATCHER -DYGZ_REF_FRAME -DYGZ_BOUNDARY_BUILD -DYGZ_REAL_DBOW2 -DYGZF_WITH_REFERENCE_HEADERS"
INC="-I$S -I$ROOT/oracle -I$REF/include -I$REF -I$H -include $S/dbow2_stubs.h -include $H/ORBextractor.h"
# The reference's own src/ORBmatcher.cc (+ src/Align.cc) stays in the link for the members outside the hot path (Fuse x2, SearchBySim3,
# SearchForTriangulation, SearchByProjection(KF, Scw, ...), SearchByBoW(KF, KF, ...)): its definitions are made WEAK, so the strong
# definitions of the hot-path members in the product's ORBmatcher.cc win at link time and nothing in the reference file is edited.
g++ $FLAGS $INC -c "$REF/src/ORBmatcher.cc" -o "$OUT/ref_ORBmatcher.o"
g++ $FLAGS $INC -c "$REF/src/Align.cc" -o "$OUT/ref_Align.o"
objcopy --weaken "$OUT/ref_ORBmatcher.o"
g++ -O2 -std=c++14 -msse4.2 -pthread -w -ffp-contract=off \
    -DYGZ_REF_MATCHER -DYGZ_REF_FRAME -DYGZ_BOUNDARY_BUILD -DYGZ_REAL_DBOW2 -DYGZF_WITH_REFERENCE_HEADERS \
    -I"$S" -I"$ROOT/oracle" -I"$REF/include" -I"$REF" -I"$H" \
    -include "$S/dbow2_stubs.h" -include "$H/ORBextractor.h" \
    "$REF/src/Frame.cc" "$OUT/ref_ORBmatcher.o" "$OUT/ref_Align.o" \
    "$H/ORBextractor.cc" "$H/ORBmatcher.cc" "$H/SparseImageAlign.cc" "$H/ORBVocabularyDevice.cc" "$H/ygzf_pool.cc" \
    "$REF/Thirdparty/DBoW2/DBoW2/FORB.cpp" "$REF/Thirdparty/DBoW2/DBoW2/BowVector.cpp" "$REF/Thirdparty/DBoW2/DBoW2/FeatureVector.cpp" \
    "$REF/Thirdparty/DBoW2/DBoW2/ScoringObject.cpp" "$REF/Thirdparty/DBoW2/DUtils/Random.cpp" "$REF/Thirdparty/DBoW2/DUtils/Timestamp.cpp" \
    "$ROOT/tests/cpp/boundary_frame.cc" "$ROOT/tests/cpp/mini_cv_nocompute.cpp" \
    "$ROOT/oracle/oracle_align.cpp" "$ROOT/oracle/oracle_direct.cpp" \
    -L"$ROOT/orb_ygz_slam_amd/lib" -lygzf -Wl,-rpath,'$ORIGIN/../../../orb_ygz_slam_amd/lib' \
    -o "$OUT/boundary_frame"
rm -f "$OUT/ref_ORBmatcher.o" "$OUT/ref_Align.o"
# strong (T) = the product's definition was linked; weak (W) = the reference's body is still the one in use
nm -C "$OUT/boundary_frame" | grep -E " [TW] ygz::ORBmatcher::(SearchByProjection|SearchByBoW|SearchForInitialization|FindDirectProjection|Fuse|SearchBySim3|SearchForTriangulation|DescriptorDistance)\(" | sed 's/^[0-9a-f]* //' | sort > "$OUT/boundary_frame.symbols"
echo "built $OUT/boundary_frame"
