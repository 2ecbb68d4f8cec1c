#!/bin/bash
# after the hand-over, the dealing and the refitted crossover: the whole GPU suite (defaults), the tree / queue table, the suite with the hand-over forced
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05y
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 -m pytest tests -m "gpu and not slow" -x -q -rs > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/gpu_tests.log | cut -c1-200
timeout -k 10 400 python3 tools/tree_vs_queue.py 2>&1 | grep -v amdgpu.ids | tee $OUT/tree_vs_queue.txt
PTRACE_QCHOICE=2 PTRACE_Q_FEW_LANES=64 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m "gpu and not slow" -x -q > $OUT/tests_few64.log 2>&1; echo "pytest (queue forced, every pixel handed over once the queue is dry) rc $?"; tail -2 $OUT/tests_few64.log | cut -c1-200
