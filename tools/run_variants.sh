#!/bin/bash
# usage (GPU box): VARIANTS="base occ6 occ8" tools/run_variants.sh [bench args]: kernel times + images/s per prebuilt
# library of instance_stixels_amd/lib/variants (tools/build_variant.sh); "base" = the product library
set -u
export TMPDIR=/tmp
for v in ${VARIANTS:-base}; do
  echo "== $v"; mkdir -p gpurun_out; echo "$(date +%T) $v" >> gpurun_out/run_variants.progress
  ( if [ "$v" != base ]; then export IS_CORE_LIB=$PWD/instance_stixels_amd/lib/variants/libis_core_$v.so; fi
    rm -rf /tmp/prof_var; timeout -k 10 ${VAR_TIMEOUT:-150} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_var -- python3 bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-single --no-d2h --no-verify --no-prune-stats --min-seconds 0 "$@" > /tmp/var.log 2>&1 )
  python3 - <<'PY'
import csv,glob,json
f=glob.glob('/tmp/prof_var/**/*kernel_stats.csv',recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if float(r['Percentage'])>3:
            print('  ', r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
try:
    d=json.loads([l for l in open('/tmp/var.log') if l.startswith('{')][-1]); print('   images/s', round(d['value']))
except Exception as e: print('   no bench line', e)
PY
done
