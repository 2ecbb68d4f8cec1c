#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05l
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests -m "gpu and not slow" -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/gpu_tests.log
for rep in 1 2 3 4; do timeout -k 10 200 python3 tools/kbench.py c3n10 --rounds 12 2>/dev/null | cut -c1-100; done | tee $OUT/kbench_tree_commit.txt
PTRACE_QCHOICE=0 timeout -k 10 200 python3 tools/kbench.py demo10 c2n10 --rounds 4 2>/dev/null | cut -c1-100 | tee -a $OUT/kbench_tree_commit.txt
PT_FUZZ_SEEDS=200 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k random_scenes > $OUT/fuzz200.log 2>&1; echo "fuzz rc $?"; tail -2 $OUT/fuzz200.log
