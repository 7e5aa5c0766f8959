#!/bin/bash
# GPU box: kernel trace of one pairwise bench run; per-tile durations of k_pw_phase1 / k_pw_phase2 (last step).
# usage: bash tools/trace_tiles.sh [bench args]
set -u
ARGS=${@:---preset drn_d_38_pairwise --steps 2 --warmup 1 --no-cpu-baseline --no-single --no-d2h --no-variants --no-verify --no-prune-stats --min-seconds 0}
OUT=gpurun_out/trace_tiles
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py $ARGS > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
p1 = [r for r in rows if "k_pw_phase1" in r["Kernel_Name"]]
p2 = [r for r in rows if "k_pw_phase2" in r["Kernel_Name"] and "generic" not in r["Kernel_Name"]]
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
n = 16
if p1:
    a, b = p1[-n:], p2[-n:]
    print("tile  phase1_us  phase2_us  gap_before_p1_us")
    for i in range(n):
        gap = (int(a[i]["Start_Timestamp"]) - int((b[i-1] if i else a[i])["End_Timestamp"])) / 1e3 if i else 0
        print(f"{i:4d} {d(a[i]):10.1f} {d(b[i]):10.1f} {gap:8.1f}")
    print("sum", round(sum(map(d, a)), 1), round(sum(map(d, b)), 1), "span", (int(b[-1]["End_Timestamp"]) - int(a[0]["Start_Timestamp"])) / 1e3)
agg = collections.defaultdict(list)
for r in rows: agg[r["Kernel_Name"][:60]].append(d(r))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{k:60s} n={len(v):4d} avg={sum(v)/len(v):9.1f} us total={sum(v)/1e3:8.2f} ms")
PY
