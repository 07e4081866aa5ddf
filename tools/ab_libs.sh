#!/bin/bash
# tools/ab_libs.sh <git-ref> -- builds libygzf.so of another commit into orb_ygz_slam_amd/lib_ab/libygzf_<ref>.so (run HERE, before gpurun) so that
# one GPU-box session can time two builds side by side: box-to-box variation (~2 %) is larger than most kernel changes.
#   bash tools/ab_libs.sh HEAD~1
#   gpurun -- 'YGZF_LIBRARY=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_HEAD~1.so python bench.py ...; python bench.py ...'
set -e
REF=${1:?git ref}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" worktree add -q --detach "$TMP/w" "$REF"
(cd "$TMP/w" && python orb_ygz_slam_amd/build.py --force > /dev/null)
mkdir -p "$ROOT/orb_ygz_slam_amd/lib_ab"
cp "$TMP/w/orb_ygz_slam_amd/lib/libygzf.so" "$ROOT/orb_ygz_slam_amd/lib_ab/libygzf_$REF.so"
git -C "$ROOT" worktree remove --force "$TMP/w"
rm -rf "$TMP"
echo "$ROOT/orb_ygz_slam_amd/lib_ab/libygzf_$REF.so"
