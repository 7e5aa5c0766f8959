# GPU box: quick A/B figures of the default library and of ablation builds lib/libis_core_abl<X>.so
# (make -C instance_stixels_amd/csrc abl ABL="-D..." ; cp lib/libis_core_abl.so lib/libis_core_ablX.so).
#   BENCH_ARGS="--preset drn_d_38_pairwise" bash tools/unq.sh A B      ("" = lib/libis_core_abl.so)
# prints: library, frames/s, DP ms per step, single-frame ms, --verify result
B="timeout -k 10 300 python bench.py $BENCH_ARGS --no-variants --no-cpu-baseline --no-d2h --verify"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), d["kernel_ms"]["dp_ms"], d.get("single_frame",{}).get("ms_per_frame"), d.get("verify",{}).get("ok"))'
$B | python -c "$P" default
for L in "$@"; do IS_CORE_LIB=instance_stixels_amd/lib/libis_core_abl$L.so $B | python -c "$P" abl$L; done
