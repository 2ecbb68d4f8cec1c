#!/bin/bash
# with the hand-over in place: where the one-queue kernel's frame stack lives (1 LDS: one workgroup per CU; 2 split, 0 HBM: two)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05ai
mkdir -p $OUT
cd $ROOT
for h in 1 2 0; do
  echo "== PTRACE_Q_FRAMES_HOME=$h"
  PTRACE_Q_FRAMES_HOME=$h timeout -k 10 200 python3 tools/kbench.py c2n10 demo10 --rounds 10 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee $OUT/home.txt
