#!/bin/bash
# what two waves per SIMD give the one-queue kernel when its frames stay in LDS: C2 + plane N = 10 at D = 1 (40 KB of frames per workgroup)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05o
mkdir -p $OUT
cd $ROOT
export PTRACE_QCHOICE=2
for d in 1 2; do
for wg in 1 2 3; do
  echo "== D=$d one-queue kernel forced, workgroups per CU $wg"
  KB_DEPTH=$d PTRACE_Q_WG_PER_CU=$wg timeout -k 10 120 python3 tools/kbench.py c2n10 demo10 --rounds 6 2>&1 | grep -v amdgpu.ids | cut -c1-110
done; done | tee $OUT/q_waves.txt
