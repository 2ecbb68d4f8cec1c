#!/bin/bash
# round 5: the N-rank path rehearsed on the one-GPU box (two gloo ranks sharing the card): the device side of the gather, bench --gpus 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05g
mkdir -p $OUT
cd $ROOT
echo skip tests
PT_DIST_BACKEND=gloo PT_BENCH_ORACLE_S=${ORACLE_S:-30} timeout -k 10 500 python3 bench.py --gpus 2 --steps 2 --warmup 2 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "bench --gpus 2 (gloo) rc $?"; tail -c 3000 $OUT/bench_gloo2.err; head -c 6000 $OUT/bench_gloo2.json
