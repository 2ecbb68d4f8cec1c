#!/bin/bash
# the one-queue kernel hands pixels over budget to the tree kernel: parity with the hand-over forced on many pixels, then the budget swept
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05p
mkdir -p $OUT
cd $ROOT
PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=15 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m "gpu and not slow" -x -q > $OUT/tests_budget15.log 2>&1; echo "pytest (queue forced, budget 15) rc $?"; tail -3 $OUT/tests_budget15.log | cut -c1-200
for b in 0 100 200 300 400 600 800; do
  echo "== budget $b"
  PTRACE_Q_BUDGET=$b timeout -k 10 120 python3 tools/kbench.py c2n10 demo10 c3n10 --rounds 6 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee $OUT/budget_sweep.txt
echo "== c3n10 one-queue forced"
for b in 0 200 400; do
  echo "== budget $b"
  PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=$b timeout -k 10 120 python3 tools/kbench.py c3n10 --rounds 6 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee -a $OUT/budget_sweep.txt
