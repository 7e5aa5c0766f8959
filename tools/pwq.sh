# GPU box: quick pairwise figures; IS_P2_SPLIT=1 selects the split phase-2 kernel
B="timeout -k 10 300 python bench.py --preset drn_d_38_pairwise --no-variants --no-cpu-baseline --no-d2h"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), d["kernel_ms"]["dp_ms"], d.get("single_frame",{}).get("ms_per_frame"), d.get("verify"))'
$B --verify | python -c "$P" default
IS_P2_SPLIT=1 $B --verify | python -c "$P" split
