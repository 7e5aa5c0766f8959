#!/bin/bash
# usage (GPU box): ENVS="IS_LUT_CARRY=0 IS_LUT_CARRY=1" tools/ab_env.sh [bench args]: kernel times + images/s of the product
# library under each environment setting (the IS_* knobs are read when a context is created), same box, same run
set -u
export TMPDIR=/tmp
for e in ${ENVS:-X=0}; do
  echo "== $e"; mkdir -p gpurun_out; echo "$(date +%T) $e" >> gpurun_out/run_variants.progress
  ( export "$e"
    rm -rf /tmp/prof_var; timeout -k 10 ${VAR_TIMEOUT:-150} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_var -- python3 bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-single --no-d2h --no-verify --no-prune-stats --min-seconds 0 "$@" > /tmp/var.log 2>&1 )
  python3 - <<'PY'
import csv,glob,json
f=glob.glob('/tmp/prof_var/**/*kernel_stats.csv',recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if float(r['Percentage'])>3:
            print('  ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
try:
    d=json.loads([l for l in open('/tmp/var.log') if l.startswith('{')][-1]); print('   images/s', round(d['value']))
except Exception as e: print('   no bench line', e); print(open('/tmp/var.log').read()[-1500:])
PY
done
