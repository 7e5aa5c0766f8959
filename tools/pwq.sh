# GPU box: quick pairwise figures of the default library and of lib/libis_core_abl.so
B="python bench.py --preset drn_d_38_pairwise --no-variants --no-cpu-baseline --no-d2h"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), d["kernel_ms"]["dp_ms"], d.get("single_frame",{}).get("ms_per_frame"), d.get("verify"))'
$B --verify | python -c "$P" default
IS_CORE_LIB=instance_stixels_amd/lib/libis_core_abl.so $B --verify | python -c "$P" abl
