#!/bin/bash
# C3 N = 10: tree kernel against one-queue kernel + hand-over at small budgets, 24 frames each, twice
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05ae
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
  echo "== tree"; PTRACE_QCHOICE=0 timeout -k 10 100 python3 tools/kbench.py c3n10 --rounds 24 2>&1 | grep -v amdgpu.ids | cut -c1-110
  for b in 12 14 20; do
    echo "== queue forced, budget $b"; PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=$b timeout -k 10 100 python3 tools/kbench.py c3n10 --rounds 24 2>&1 | grep -v amdgpu.ids | cut -c1-110
  done
done | tee $OUT/c3.txt
for c in "c3 1920 1080 10 3 1" "c3 1280 720 3 5 1" "plane 320 180 3 5 1" "demo 320 240 10 3 1" "c3 640 360 10 3 1"; do
  for q in 0 2; do echo -n "$c QCHOICE=$q budget 12: "; PTRACE_QCHOICE=$q PTRACE_Q_BUDGET=12 timeout -k 10 120 python3 tools/tree_vs_queue.py --one $c 2>&1 | grep -v amdgpu.ids; done
done | tee -a $OUT/c3.txt
