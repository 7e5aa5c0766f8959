#!/bin/bash
# GPU box: average duration of the pairwise kernels at two batch sizes (round quantisation of k_pw_phase2x).
export TMPDIR=/tmp
for B in ${@:-56 64}; do
  rm -rf /tmp/pb$B
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb$B -- python3 /root/repo/bench.py --preset drn_d_38_pairwise --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-single --no-d2h --no-verify --no-prune-stats --min-seconds 0 > /tmp/pb$B.log 2>&1)
  python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/pb$B/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_pw_phase" in r["Name"]:
            print("batch $B", r["Name"][:28], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
done
