#!/bin/bash
# A/B builds of libhmme.so with extra compile flags, next to the real one: hm-opencl_amd/csrc/build/variants/libhmme_<tag>.so.
# Run a variant with HMME_LIB=<path> python bench.py ...  (hmme/api.py honours HMME_LIB).  usage: tools/build_variant.sh <tag> <flags...>
set -e
TAG=$1; shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=$HERE/hm-opencl_amd/csrc/build/variants
mkdir -p $OUT
cd $OUT
# the same build id recipe as hm-opencl_amd/csrc/Makefile (sources + arch + the variant's flags): hmme_build_id() of a variant is its own,
# so that a counter summary taken on a variant can never pass for the default library's (bench.py pmc_profile)
C=$HERE/hm-opencl_amd/csrc
ID=$( (cat $C/hmme.hip $C/me_kernels.hpp $C/me_tree_fen0.inc $C/me_tree_fen1.inc $C/me_tree16_fen0.inc $C/me_tree16_fen1.inc $C/me_slotmap.inc; echo "gfx950 $*") | sha256sum | cut -c1-16)
# -save-temps like the Makefile's compile: the phased pipeline it switches on does not generate the same code as the one-step compile (the
# one-step build of me_search_kernel<1, 0> comes out at 256 VGPRs with spills and fails tools/check_dpp_hazard.py) -- a variant must differ
# from the default library in its -D flags only.  Temporaries in a directory of the variant's own, the ISA kept beside the library.
mkdir -p tmp_$TAG && cd tmp_$TAG
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -DHMME_BUILD_ID=\"$ID\" -c -save-temps $HERE/hm-opencl_amd/csrc/hmme.hip -o ../hmme_$TAG.o
mv hmme-hip-amdgcn-amd-amdhsa-gfx950.s ../libhmme_$TAG.s
cd .. && rm -rf tmp_$TAG
python3 $HERE/tools/check_dpp_hazard.py libhmme_$TAG.s > /dev/null || { echo "variant $TAG fails tools/check_dpp_hazard.py" >&2; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o libhmme_$TAG.so hmme_$TAG.o
echo $OUT/libhmme_$TAG.so
