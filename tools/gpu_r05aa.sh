#!/bin/bash
# budget from the flagged pixels (default): tail budget x few lanes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05aa
mkdir -p $OUT
cd $ROOT
for cfg in "50 16" "0 16" "25 16" "100 16" "50 8" "50 24" "50 32" "100 24" "25 8" "0 0"; do
  set -- $cfg
  echo "== budget by the device, tail budget $1, few lanes $2"
  PTRACE_Q_TAIL_BUDGET=$1 PTRACE_Q_FEW_LANES=$2 timeout -k 10 120 python3 tools/kbench.py c2n10 demo10 --rounds 8 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee $OUT/sweep3.txt
