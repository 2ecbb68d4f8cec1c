#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05k
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests -m "gpu and not slow" -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/gpu_tests.log
for rep in 1 2 3; do for v in 1 0; do echo "== PTRACE_TREE_JUMP=$v"; PTRACE_TREE_JUMP=$v timeout -k 10 200 python3 tools/kbench.py c3n10 --rounds 12 2>/dev/null | cut -c1-100; done; done | tee $OUT/kbench_tree_jump.txt
