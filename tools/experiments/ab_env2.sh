#!/bin/bash
# usage (GPU box): tools/ab_env2.sh "A=1 B=2" "A=3" -- [bench args]: one bench.py run (no profiler) per environment SET
# (space-separated assignments in one quoted argument; "-" = no setting): images/s and the bench's own kernel_ms
set -u
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
[ $# -gt 0 ] && shift
mkdir -p gpurun_out
for e in "${sets[@]}"; do
  echo "== $e"; echo "$(date +%T) $e" >> gpurun_out/run_variants.progress
  ( if [ "$e" != "-" ]; then export $e; fi
    timeout -k 10 ${VAR_TIMEOUT:-200} python3 bench.py --batch 64 --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-single --no-d2h --no-prune-stats --min-seconds 1 "$@" > /tmp/var.log 2>&1 )
  python3 - <<'PY'
import json
try:
    d=json.loads([l for l in open('/tmp/var.log') if l.startswith('{')][-1]); print('   images/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms'].items()}, 'verify', d.get('verify',{}).get('ok'))
except Exception as e: print('   no bench line', e); print(open('/tmp/var.log').read()[-1500:])
PY
done
