#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace + PMC passes of bench.py, CSVs into gpurun_out/.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
# (the driver's step counts: the first dispatches of a process run 5-15 % slower -- clocks, cold caches -- and 25 of them
# dilute that in the per-kernel averages, which then agree with bench.py's HIP-event kernel_ms)
ARGS=${@:---batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-single --no-d2h --no-verify --no-prune-stats --min-seconds 0}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
echo "python3 bench.py $ARGS" > $OUT/command.txt
export TMPDIR=/tmp
export IS_PW_GROUPS=1   # one column group: per-kernel durations of launches that do not overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc1 -- python3 bench.py $ARGS > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc2 -- python3 bench.py $ARGS > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py $ARGS > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- python3 bench.py $ARGS > $OUT/bench_pmc4.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_TRANS SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc5 -- python3 bench.py $ARGS > $OUT/bench_pmc5.log 2>&1
find $OUT -name "*.csv" | head -40
du -sh $OUT
