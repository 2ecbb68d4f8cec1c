#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05m
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests -m "gpu and not slow" -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/gpu_tests.log
PTRACE_QCHOICE=2 timeout -k 10 600 python3 -m pytest tests -m "gpu and not slow" -q > $OUT/gpu_tests_queue.log 2>&1; echo "pytest (queue forced) rc $?"; tail -2 $OUT/gpu_tests_queue.log
for rep in 1 2 3; do timeout -k 10 300 python3 tools/kbench.py c3n10 c2n10 demo10 c3 c3:sample c4:sample c2s2 pl --rounds 10 2>/dev/null | cut -c1-100; done | tee $OUT/kbench.txt
PT_FUZZ_SEEDS=120 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k random_scenes > $OUT/fuzz200.log 2>&1; echo "fuzz rc $?"; tail -2 $OUT/fuzz200.log
PTRACE_QCHOICE=2 PT_FUZZ_SEEDS=60 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k random_scenes > $OUT/fuzz150_queue.log 2>&1; echo "fuzz (queue forced) rc $?"; tail -2 $OUT/fuzz150_queue.log
