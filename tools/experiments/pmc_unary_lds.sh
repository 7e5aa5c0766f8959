export TMPDIR=/tmp
ARGS="--batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-single --no-d2h --no-verify --no-prune-stats --min-seconds 0"
rm -rf /tmp/p1 /tmp/p2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 bench.py $ARGS > /tmp/l1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/p2 -- python3 bench.py $ARGS > /tmp/l2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/p1/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_dp_unary_fast' in r['Name'] or 'prepare' in r['Name']: print(r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
agg=collections.defaultdict(float);cnt=collections.defaultdict(int)
for f in glob.glob('/tmp/p2/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_dp_unary_fast' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value']);cnt[r['Counter_Name']]+=1
for k in agg: print(k, agg[k]/cnt[k])
print('conflict/active', agg['SQ_LDS_BANK_CONFLICT']/agg['SQ_LDS_IDX_ACTIVE'])
PY
