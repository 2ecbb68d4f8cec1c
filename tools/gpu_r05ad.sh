#!/bin/bash
# C3 N = 10 through the one-queue kernel with small budgets (every pixel starts at once; the heavy ones go to the tree kernel early)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05ad
mkdir -p $OUT
cd $ROOT
for b in 8 12 16 24 32 48; do
  echo "== queue forced, budget $b"
  PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=$b timeout -k 10 100 python3 tools/kbench.py c3n10 --rounds 16 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee $OUT/c3_budget.txt
echo "== tree"; PTRACE_QCHOICE=0 timeout -k 10 100 python3 tools/kbench.py c3n10 --rounds 16 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee -a $OUT/c3_budget.txt
