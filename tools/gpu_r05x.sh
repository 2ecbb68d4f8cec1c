#!/bin/bash
# the one-queue kernel deals out flagged pixels (units) instead of pixel indices; 4K frames with num_of_rays > 1 under the three choices
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05x
mkdir -p $OUT
cd $ROOT
PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=5 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m "gpu and not slow" -x -q > $OUT/tests_budget5.log 2>&1; echo "pytest (queue forced, budget 5) rc $?"; tail -2 $OUT/tests_budget5.log | cut -c1-200
timeout -k 10 200 python3 tools/kbench.py c3n10 c2n10 demo10 --rounds 8 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee $OUT/kbench.txt
for q in 0 2 1; do
  echo "== PTRACE_QCHOICE=$q"
  for c in "c3 1280 720 10 3 1" "c3 1920 1080 10 3 1" "c3 3840 2160 10 3 1" "demo 1920 1440 10 3 1" "plane 3840 2160 10 3 1" "demo 320 240 10 3 1" "plane 640 360 10 3 1"; do
    echo -n "$c: "; PTRACE_QCHOICE=$q timeout -k 10 120 python3 tools/tree_vs_queue.py --one $c 2>&1 | grep -v amdgpu.ids
  done
done | tee $OUT/big_frames.txt
