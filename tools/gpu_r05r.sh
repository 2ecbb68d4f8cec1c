#!/bin/bash
# hand-over to the tree kernel once the pixel queue has run dry: the tail budget swept (fixed budget off), and combined
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05r
mkdir -p $OUT
cd $ROOT
for t in 0 10 25 50 100 200; do
  echo "== budget 0, tail budget $t"
  PTRACE_Q_BUDGET=0 PTRACE_Q_TAIL_BUDGET=$t timeout -k 10 120 python3 tools/kbench.py c2n10 demo10 --rounds 6 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee $OUT/tail_sweep.txt
for cfg in "400 50" "600 50" "800 50" "800 25"; do
  set -- $cfg
  echo "== budget $1, tail budget $2"
  PTRACE_Q_BUDGET=$1 PTRACE_Q_TAIL_BUDGET=$2 timeout -k 10 120 python3 tools/kbench.py c2n10 demo10 --rounds 6 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee -a $OUT/tail_sweep.txt
PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=0 PTRACE_Q_TAIL_BUDGET=3 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m "gpu and not slow" -x -q > $OUT/tests_tail3.log 2>&1; echo "pytest (queue forced, tail budget 3) rc $?"; tail -3 $OUT/tests_tail3.log | cut -c1-200
