#!/bin/bash
# development build of the library with per-wave cycle stamps in k_pyrlk and the homography kernels (tools/lk_stamps.py, tools/hg_stamps.py read them)
set -e
cd "$(dirname "$0")/../visualslam_amd/csrc"
mkdir -p ../../tools/bin
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -I../../include -I. -DVSA_LK_STAMPS -DVSA_HG_STAMPS -DVSA_MG_STAMPS -fno-slp-vectorize -shared -x hip image_ops.hip corner_track.hip pyrlk.hip homography.hip nvxcu_api.cpp ovx_*.cpp \
  -o ../../tools/bin/libvisualslam_amd_stamps.so
