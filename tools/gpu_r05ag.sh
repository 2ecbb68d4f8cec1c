#!/bin/bash
# random scenes with five rays per hit: the one-queue kernel forced and handing over after 6 rays / when the queue runs dry; and at the defaults
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05ag
mkdir -p $OUT
cd $ROOT
PT_FUZZ_SEEDS=160 PT_FUZZ_RAYS=5 PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=6 timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k random_scenes > $OUT/fuzz160_n5_budget6.log 2>&1; echo "fuzz N=5 (queue forced, budget 6) rc $?"; tail -2 $OUT/fuzz160_n5_budget6.log
PT_FUZZ_SEEDS=160 PT_FUZZ_RAYS=5 PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=0 PTRACE_Q_TAIL_BUDGET=0 PTRACE_Q_FEW_LANES=64 timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k random_scenes > $OUT/fuzz160_n5_few64.log 2>&1; echo "fuzz N=5 (queue forced, all in flight handed over) rc $?"; tail -2 $OUT/fuzz160_n5_few64.log
PT_FUZZ_SEEDS=160 PT_FUZZ_RAYS=5 timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k random_scenes > $OUT/fuzz160_n5.log 2>&1; echo "fuzz N=5 (defaults) rc $?"; tail -2 $OUT/fuzz160_n5.log
