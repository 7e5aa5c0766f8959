#!/bin/bash
# usage: ENV_VARIANTS="base IS_X=1 IS_X=2,IS_Y=0" tools/abl_env.sh [bench args]  (GPU box): kernel times and
# images/s of the in-tree library under run-time knobs (environment variables read at is_ctx_create)
set -u
export TMPDIR=/tmp
for v in ${ENV_VARIANTS:-base}; do
  echo "== $v"
  ( if [ "$v" != base ]; then for kv in ${v//,/ }; do export "$kv"; done; fi
    rm -rf /tmp/prof_env; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_env -- python3 bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-single --no-d2h --no-verify --no-prune-stats --min-seconds 0 "$@" > /tmp/env.log 2>&1 )
  python3 - <<'PY'
import csv,glob,json
f=glob.glob('/tmp/prof_env/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>3:
        print('  ', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
try:
    d=json.loads([l for l in open('/tmp/env.log') if l.startswith('{')][-1]); print('   images/s', round(d['value']), 'prepare_ms', round(d['kernel_ms']['prepare_ms'],3))
except Exception as e: print('   no bench line', e)
PY
done
