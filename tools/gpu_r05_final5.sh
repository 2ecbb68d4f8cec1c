#!/bin/bash
# round 5, final evidence, part 2 on the final binary: the N-rank bench rehearsed under gloo (oracle check of the gathered frames), the suite with forced
# kernel variants (one queue, round-2 kernels, tree kernel), 600 random scenes at the defaults
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_final
mkdir -p $OUT
cd $ROOT
PT_DIST_BACKEND=gloo PT_BENCH_ORACLE_S=400 timeout -k 10 600 python3 bench.py --gpus 2 --steps 4 --warmup 2 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "bench --gpus 2 (gloo) rc $?"; grep "^\[bench" $OUT/bench_gloo2.err | tail -4
PTRACE_QCHOICE=2 timeout -k 10 400 python3 -m pytest tests -m "gpu and not slow" -q > $OUT/gpu_tests_queue_forced.log 2>&1; echo "pytest (one-queue forced) rc $?"; tail -2 $OUT/gpu_tests_queue_forced.log
PTRACE_QCHOICE=0 timeout -k 10 400 python3 -m pytest tests -m "gpu and not slow" -q > $OUT/gpu_tests_tree_forced.log 2>&1; echo "pytest (tree kernel forced) rc $?"; tail -2 $OUT/gpu_tests_tree_forced.log
PTRACE_TREE=0 PTRACE_TILE4=0 timeout -k 10 400 python3 -m pytest tests -m "gpu and not slow" -q > $OUT/gpu_tests_round2_kernels_forced.log 2>&1; echo "pytest (round-2 kernels forced) rc $?"; tail -2 $OUT/gpu_tests_round2_kernels_forced.log
PT_FUZZ_SEEDS=600 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k random_scenes > $OUT/fuzz600.log 2>&1; echo "fuzz rc $?"; tail -2 $OUT/fuzz600.log
