#!/bin/bash
# usage: build_variant.sh <git-rev> <name> -- builds the library as it was at <git-rev> into tools/bin/libvisualslam_amd_<name>.so
# (for tools/exp_lib_ab.sh: A/B of library builds on the bench)
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" visualslam_amd/csrc include | tar -x -C "$TMP"
cd "$TMP/visualslam_amd/csrc"
mkdir -p "$ROOT/tools/bin"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
  -I"$TMP/include" -I. -shared -x hip image_ops.hip corner_track.hip pyrlk.hip homography.hip nvxcu_api.cpp ovx_*.cpp \
  -o "$ROOT/tools/bin/libvisualslam_amd_$NAME.so"
rm -rf "$TMP"
ls -la "$ROOT/tools/bin/libvisualslam_amd_$NAME.so"
