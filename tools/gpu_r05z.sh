#!/bin/bash
# the new full-size / hand-over tests, then random scenes with the hand-over forced on nearly every pixel, then at the defaults
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05z
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 -m pytest tests/test_gpu_fullsize.py -m "gpu and not slow" -x -q -s -k "handed or cli_defaults" > $OUT/new_tests.log 2>&1; echo "pytest rc $?"; grep -E "handed to|passed|failed|Error" $OUT/new_tests.log | cut -c1-220 | tail -12
PT_FUZZ_SEEDS=300 PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=3 timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k random_scenes > $OUT/fuzz300_handover.log 2>&1; echo "fuzz (queue forced, budget 3) rc $?"; tail -2 $OUT/fuzz300_handover.log
PT_FUZZ_SEEDS=150 PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=0 PTRACE_Q_TAIL_BUDGET=0 PTRACE_Q_FEW_LANES=64 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k random_scenes > $OUT/fuzz150_few64.log 2>&1; echo "fuzz (queue forced, every pixel in flight handed over) rc $?"; tail -2 $OUT/fuzz150_few64.log
PT_FUZZ_SEEDS=150 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k random_scenes > $OUT/fuzz150.log 2>&1; echo "fuzz (defaults) rc $?"; tail -2 $OUT/fuzz150.log
