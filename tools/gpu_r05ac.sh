#!/bin/bash
# the one-queue kernel with a mate (frames in HBM) on the second wave slot of every SIMD: times, then parity with the queue forced
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05ac
mkdir -p $OUT
cd $ROOT
for m in 0 1; do
  echo "== PTRACE_Q_MATE=$m"
  PTRACE_Q_MATE=$m timeout -k 10 200 python3 tools/kbench.py c2n10 demo10 c3n10 --rounds 10 2>&1 | grep -v amdgpu.ids | cut -c1-110
  PTRACE_Q_MATE=$m PTRACE_QCHOICE=2 timeout -k 10 200 python3 tools/kbench.py c3n10 --rounds 10 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee $OUT/mate.txt
PTRACE_QCHOICE=2 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m "gpu and not slow" -x -q > $OUT/tests_queue.log 2>&1; echo "pytest (queue forced, mate on) rc $?"; tail -2 $OUT/tests_queue.log | cut -c1-200
