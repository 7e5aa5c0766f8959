#!/bin/bash
# usage (HERE, not on the GPU box): tools/build_variant.sh <name> <files> <flags...>
#   e.g. tools/build_variant.sh occ6 is_k_unary_fast -DISF_OCC=6
# Builds instance_stixels_amd/lib/variants/libis_core_<name>.so: the product objects, with the listed kernel
# files (comma separated, without .hip; "all" = everything) recompiled with the flags -- through the same
# rules as the product build, INCLUDING the hidden-request ISA check.  Ships to the GPU box with gpurun.
set -eu
name=$1; files=$2; shift 2
cd "$(dirname "$0")/../instance_stixels_amd/csrc"
make -j8 > /dev/null
rm -rf build_abl && cp -r build build_abl
if [ "$files" = all ]; then rm -f build_abl/*.o; else for f in ${files//,/ }; do rm -f build_abl/$f.o; done; fi
rm -f build_abl/srec.ok
make -j8 BUILD=$PWD/build_abl OUT=$PWD/../lib/abl ABL="$*" all 2>&1 | grep -E "error|violations|checked" || true
mkdir -p ../lib/variants && cp ../lib/abl/libis_core.so ../lib/variants/libis_core_$name.so && echo "built variants/libis_core_$name.so"
