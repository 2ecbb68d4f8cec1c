#!/bin/bash
# budget never below N + 2, crossover at min(R, 80): the table again, C3 N = 10 at the defaults
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05af
mkdir -p $OUT
cd $ROOT
timeout -k 10 200 python3 tools/kbench.py c3n10 c2n10 demo10 --rounds 16 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee $OUT/kbench.txt
timeout -k 10 500 python3 tools/tree_vs_queue.py 2>&1 | grep -v amdgpu.ids | tee $OUT/tree_vs_queue.txt
for c in "c3 640 360 10 3 1" "c3 960 540 10 3 1" "c3 3840 2160 10 3 1"; do
  for q in 0 2 1; do echo -n "$c QCHOICE=$q: "; PTRACE_QCHOICE=$q timeout -k 10 120 python3 tools/tree_vs_queue.py --one $c 2>&1 | grep -v amdgpu.ids; done
done | tee -a $OUT/tree_vs_queue.txt
