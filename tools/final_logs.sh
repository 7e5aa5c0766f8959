#!/bin/bash
# GPU box: what is kept under profiles/ per round (written to gpurun_out/final/): the driver's command for both
# models (compact line + complete object), the --full sweep of the headline model, the one-rank RCCL logs.
# usage: tools/final_logs.sh [full]     ("full": also the --full sweep, about two minutes)
set -u
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --out $OUT/bench_unary_batch64_full.json ) > $OUT/bench_unary_batch64.json 2> $OUT/bench_unary.err
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --preset drn_d_38_pairwise --out $OUT/bench_pairwise_batch64_full.json ) > $OUT/bench_pairwise_batch64.json 2> $OUT/bench_pairwise.err
wc -c $OUT/bench_unary_batch64.json $OUT/bench_pairwise_batch64.json
if [ "${1:-}" = full ]; then
  python3 bench.py --full --steps 5 --warmup 2 --out $OUT/bench_unary_batch64_sweep.json > $OUT/bench_unary_sweep_line.json 2> $OUT/bench_unary_sweep.err
  tail -c 600 $OUT/bench_unary_sweep_line.json
fi
for P in unary pairwise; do
  PRESET=drn_d_22_unary; [ $P = pairwise ] && PRESET=drn_d_38_pairwise
  for G in compact fixed; do
    NCCL_DEBUG=VERSION MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
      python3 bench.py --preset $PRESET --force-dist --gather $G --steps 5 --warmup 2 --no-variants --no-cpu-baseline --no-single \
      --out $OUT/force_dist_${P}_n1_${G}_full.json > $OUT/force_dist_${P}_n1_${G}.log 2>&1
    tail -1 $OUT/force_dist_${P}_n1_${G}.log | cut -c1-200
  done
done
