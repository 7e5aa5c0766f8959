#!/bin/bash
# NOTE (round 4): `make abl` now goes through the per-object rules + ISA check and takes minutes; on the GPU box prefer
# variant libraries built beforehand with tools/build_variant.sh and run with tools/run_variants.sh (a silent build of
# more than 7 minutes is killed by gpurun).
# usage: abl_prep.sh  (run on GPU box): builds variants and times k_prepare_columns / k_object_lut separately
set -u
export TMPDIR=/tmp
cd instance_stixels_amd/csrc
for v in ${ABL_VARIANTS:-base "-DPREP_STORE_LATE=0"}; do
  if [ "$v" = base ]; then A=""; else A="$v"; fi
  make abl ABL="$A" > /dev/null 2>&1 || { echo build failed $v; continue; }
  cd ../..
  rm -rf /tmp/prof_abl; IS_CORE_LIB=$PWD/instance_stixels_amd/lib/libis_core_abl.so IS_PREPARE_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_abl -- python3 bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-single --no-d2h --no-verify --no-prune-stats --min-seconds 0 > /tmp/abl.log 2>&1
  echo "== $v"; python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_abl/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'prepare' in r['Name'] or 'object_lut' in r['Name']:
        print('  ', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
  cd instance_stixels_amd/csrc
done
