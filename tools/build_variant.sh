#!/bin/bash
# Development tool: a second build of the runtime library with extra compiler flags, for A/B runs on one GPU box (HNB_LIB=<path>).
# Usage: tools/build_variant.sh <tag> <flags...>   ->  bevy_hanabi_amd/libhanabi_amd_<tag>.so
TAG=$1; shift
cd "$(dirname "$0")/../bevy_hanabi_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fPIC -shared -Wno-unused-value "$@" hanabi_amd.hip -o ../libhanabi_amd_$TAG.so -lhiprtc -ldl && echo built libhanabi_amd_$TAG.so
