#!/bin/bash
# hand-over to the tree kernel when a wave of the dry queue holds few pixels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s
mkdir -p $OUT
cd $ROOT
for f in 0 1 2 4 8 16 32; do
  echo "== few lanes $f"
  PTRACE_Q_BUDGET=0 PTRACE_Q_TAIL_BUDGET=0 PTRACE_Q_FEW_LANES=$f timeout -k 10 120 python3 tools/kbench.py c2n10 demo10 --rounds 6 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee $OUT/few_sweep.txt
for cfg in "800 4" "800 8" "1000 8"; do
  set -- $cfg
  echo "== budget $1, few lanes $2"
  PTRACE_Q_BUDGET=$1 PTRACE_Q_TAIL_BUDGET=0 PTRACE_Q_FEW_LANES=$2 timeout -k 10 120 python3 tools/kbench.py c2n10 demo10 --rounds 6 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee -a $OUT/few_sweep.txt
