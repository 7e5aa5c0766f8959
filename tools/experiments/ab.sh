#!/bin/bash
# GPU box: quick figures of the default library (and of variant libraries lib/libis_core_<X>.so given as args)
# for both presets: frames/s, DP ms, pruning-off frames/s.   usage: bash tools/ab.sh [X ...]
run() { # $1 = label, $2 = lib or ""
  if [ -n "$2" ]; then export IS_CORE_LIB=$2; else unset IS_CORE_LIB; fi
  for P in drn_d_22_unary drn_d_38_pairwise; do
    timeout -k 10 300 python bench.py --preset $P --no-variants --no-cpu-baseline --no-d2h --no-single 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$P', round(d['value']), 'dp_ms', round(d['kernel_ms']['dp_ms'],3), 'prep', round(d['kernel_ms']['prepare_ms'],3), 'verify', d.get('verify',{}).get('ok'), 'evalfrac', round(d['prune']['evaluated_frac'],4))"
    IS_NO_PRUNE=1 timeout -k 10 300 python bench.py --preset $P --no-variants --no-cpu-baseline --no-d2h --no-single --no-verify --min-seconds 0.5 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$P', 'PRUNING_OFF', round(d['value']), 'dp_ms', round(d['kernel_ms']['dp_ms'],3))"
  done
}
if [ $# -eq 0 ]; then run default ""; fi
for L in "$@"; do
  if [ "$L" = "default" ]; then run default ""; else run $L instance_stixels_amd/lib/libis_core_$L.so; fi
done
