#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass of bench.py, per-kernel averages printed.
# usage: tools/pmc_quick.sh "<counters>" [bench args...]
set -u
CNT=$1; shift
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline --no-single --no-d2h}
OUT=gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT -- python3 bench.py $ARGS > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    print(k, " ".join("%s=%.4g" % (c, agg[k][c] / cnt[k][c]) for c in sorted(agg[k])))
PY
