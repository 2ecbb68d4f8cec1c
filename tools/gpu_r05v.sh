#!/bin/bash
# hand-over policies: budget x few lanes x tail budget
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05v
mkdir -p $OUT
cd $ROOT
for cfg in "200 0 16" "300 0 16" "400 0 24" "400 0 32" "300 0 32" "200 0 32" "400 50 16" "400 100 16" "300 50 16" "200 50 0" "150 0 16" "100 0 16" "250 0 24" "300 50 32"; do
  set -- $cfg
  echo "== budget $1, tail budget $2, few lanes $3"
  PTRACE_Q_BUDGET=$1 PTRACE_Q_TAIL_BUDGET=$2 PTRACE_Q_FEW_LANES=$3 timeout -k 10 120 python3 tools/kbench.py c2n10 demo10 --rounds 6 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee $OUT/sweep2.txt
