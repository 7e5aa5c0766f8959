#!/bin/bash
# NOTE (round 4): `make abl` now goes through the per-object rules + ISA check and takes minutes; on the GPU box prefer
# variant libraries built beforehand with tools/build_variant.sh and run with tools/run_variants.sh (a silent build of
# more than 7 minutes is killed by gpurun).
# usage: ABL_VARIANTS="base -DX=1 ..." tools/abl_fused.sh [bench args]  (GPU box): k_prepare_fused time and images/s per build variant
set -u
export TMPDIR=/tmp
cd instance_stixels_amd/csrc
for v in ${ABL_VARIANTS:-base}; do
  if [ "$v" = base ]; then A=""; else A="${v//,/ }"; fi
  make abl ABL="$A" > /dev/null 2>&1 || { echo build failed $v; continue; }
  cd ../..
  rm -rf /tmp/prof_abl; IS_CORE_LIB=$PWD/instance_stixels_amd/lib/libis_core_abl.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_abl -- python3 bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-single --no-d2h --no-verify --no-prune-stats --min-seconds 0 "$@" > /tmp/abl.log 2>&1
  echo "== $v"; python3 - <<'PY'
import csv,glob,json
f=glob.glob('/tmp/prof_abl/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>3:
        print('  ', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
try:
    d=json.loads([l for l in open('/tmp/abl.log') if l.startswith('{')][-1]); print('   images/s', round(d['value']))
except Exception as e: print('   no bench line', e)
PY
  cd instance_stixels_amd/csrc
done
