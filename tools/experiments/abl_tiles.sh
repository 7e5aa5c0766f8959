#!/bin/bash
# NOTE (round 4): `make abl` now goes through the per-object rules + ISA check and takes minutes; on the GPU box prefer
# variant libraries built beforehand with tools/build_variant.sh and run with tools/run_variants.sh (a silent build of
# more than 7 minutes is killed by gpurun).
# usage: ABL_VARIANTS="base -DX=1" tools/abl_tiles.sh  (GPU box): per-tile pairwise launch times per build variant
set -u
export TMPDIR=/tmp
for v in ${ABL_VARIANTS:-base}; do
  if [ "$v" = base ]; then A=""; else A="${v//,/ }"; fi
  (cd instance_stixels_amd/csrc && make abl ABL="$A" > /dev/null 2>&1) || { echo build failed $v; continue; }
  echo "== $v"
  IS_CORE_LIB=$PWD/instance_stixels_amd/lib/libis_core_abl.so bash tools/pw_tile_times.sh
  python3 -c "
import json
d=json.loads([l for l in open('/tmp/pwt.log') if l.startswith('{')][-1]); print('   images/s', round(d['value']))"
done
