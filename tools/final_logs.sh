#!/bin/bash
# GPU box: the bench lines and the one-rank RCCL logs kept under profiles/ (written to gpurun_out/final/).
set -u
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_unary_batch64.json 2> $OUT/bench_unary.err
python bench.py --preset drn_d_38_pairwise > $OUT/bench_pairwise_batch64.json 2> $OUT/bench_pairwise.err
for P in unary pairwise; do
  PRESET=drn_d_22_unary; [ $P = pairwise ] && PRESET=drn_d_38_pairwise
  for G in compact fixed; do
    NCCL_DEBUG=VERSION MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
      python bench.py --preset $PRESET --force-dist --gather $G --steps 5 --warmup 2 --no-variants --no-cpu-baseline --no-single \
      > $OUT/force_dist_${P}_n1_${G}.log 2>&1
    tail -1 $OUT/force_dist_${P}_n1_${G}.log | cut -c1-200
  done
done
