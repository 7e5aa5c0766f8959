#!/bin/bash
# per-tile durations of the pairwise launches of the last profiled step (run on the GPU box)
export TMPDIR=/tmp
rm -rf /tmp/pwt; rocprofv3 --kernel-trace --output-format csv -d /tmp/pwt -- python3 bench.py --preset drn_d_38_pairwise --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-single --no-d2h --no-verify --no-prune-stats --min-seconds 0 "$@" > /tmp/pwt.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/pwt/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def dur(name):
    return [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if name in r['Kernel_Name']]
p1=dur('k_pw_phase1'); p2=dur('k_pw_phase2x'); pr=dur('k_prepare')
print('P1 per tile (us):',[round(x) for x in p1[-16:]], 'sum', round(sum(p1[-16:])))
print('P2x per tile (us):',[round(x) for x in p2[-16:]], 'sum', round(sum(p2[-16:])))
print('prepare', [round(x) for x in pr])
PY
