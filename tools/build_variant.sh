#!/bin/bash
# A/B builds of libhmme.so with extra compile flags, next to the real one: hm-opencl_amd/csrc/build/variants/libhmme_<tag>.so.
# Run a variant with HMME_LIB=<path> python bench.py ...  (hmme/api.py honours HMME_LIB).  usage: tools/build_variant.sh <tag> <flags...>
set -e
TAG=$1; shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=$HERE/hm-opencl_amd/csrc/build/variants
mkdir -p $OUT
cd $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c $HERE/hm-opencl_amd/csrc/hmme.hip -o hmme_$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC -o libhmme_$TAG.so hmme_$TAG.o
echo $OUT/libhmme_$TAG.so
