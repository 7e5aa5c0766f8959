#!/bin/bash
# GPU box: rocprofv3 kernel stats of one bench.py run (extra args go to bench.py; env passes through).
export TMPDIR=/tmp
rm -rf /tmp/kst
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-single --no-d2h --no-verify --no-prune-stats --min-seconds 0 "$@" > /tmp/kst.log 2>&1)
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/kst/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.3:
            print("%-44s calls %4s avg_us %9.1f  %5s%%" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
