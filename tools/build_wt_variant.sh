#!/bin/bash
# usage: build_wt_variant.sh <name> [extra compiler flags...] -- builds the WORKING TREE's library with extra flags (-D switches of an
# experiment) into tools/bin/libvisualslam_amd_<name>.so (for tools/exp_lib_ab.sh)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/visualslam_amd/csrc"
mkdir -p "$ROOT/tools/bin"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
  -I"$ROOT/include" -I. "$@" -shared -x hip image_ops.hip corner_track.hip pyrlk.hip homography.hip nvxcu_api.cpp ovx_*.cpp \
  -o "$ROOT/tools/bin/libvisualslam_amd_$NAME.so"
ls -la "$ROOT/tools/bin/libvisualslam_amd_$NAME.so"
