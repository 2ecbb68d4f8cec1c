#!/bin/bash
# the leaf round's spare lanes trace the next child of the nearest ancestor with children left: parity (tree kernel forced), times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05ab
mkdir -p $OUT
cd $ROOT
PTRACE_QCHOICE=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m "gpu and not slow" -x -q > $OUT/tests_tree.log 2>&1; echo "pytest (tree forced) rc $?"; tail -2 $OUT/tests_tree.log | cut -c1-200
timeout -k 10 200 python3 tools/kbench.py c3n10 c2n10 demo10 --rounds 12 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee $OUT/kbench.txt
PTRACE_QCHOICE=2 timeout -k 10 200 python3 tools/kbench.py c3n10 --rounds 12 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee -a $OUT/kbench.txt
PTRACE_LIB=$ROOT/build_variants/libptrace_dbg.so DBG_LANES=1 timeout -k 10 120 python3 tools/dbgtree.py c3n10 2>&1 | tail -11 | head -3
