# GPU box: quick unary figures of the default library and of lib/libis_core_abl<X>.so variants
B="timeout -k 10 300 python bench.py $BENCH_ARGS --no-variants --no-cpu-baseline --no-d2h --verify"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), d["kernel_ms"]["dp_ms"], d.get("single_frame",{}).get("ms_per_frame"), d.get("verify",{}).get("ok"))'
$B | python -c "$P" default
for L in "$@"; do IS_CORE_LIB=instance_stixels_amd/lib/libis_core_abl$L.so $B | python -c "$P" abl$L; done
