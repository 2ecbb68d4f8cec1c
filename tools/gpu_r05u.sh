#!/bin/bash
# the one-queue kernel hands pixels to the tree kernel, which goes on where the lane stopped: parity with the hand-over forced, then the policies swept
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05u
mkdir -p $OUT
cd $ROOT
run_tests() {  # name, env...
  local name=$1; shift
  env PTRACE_QCHOICE=2 "$@" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m "gpu and not slow" -x -q > $OUT/tests_$name.log 2>&1
  echo "pytest ($name: $*) rc $?"; tail -2 $OUT/tests_$name.log | cut -c1-200
}
run_tests budget5 PTRACE_Q_BUDGET=5 PTRACE_Q_TAIL_BUDGET=0 PTRACE_Q_FEW_LANES=0 && \
run_tests few64 PTRACE_Q_BUDGET=0 PTRACE_Q_TAIL_BUDGET=0 PTRACE_Q_FEW_LANES=64 && \
for cfg in "0 0 0" "0 0 4" "0 0 8" "0 0 16" "0 0 32" "0 0 48" "0 0 64" "0 50 0" "0 100 0" "400 0 0" "400 0 16" "800 0 16"; do
  set -- $cfg
  echo "== budget $1, tail budget $2, few lanes $3"
  PTRACE_Q_BUDGET=$1 PTRACE_Q_TAIL_BUDGET=$2 PTRACE_Q_FEW_LANES=$3 timeout -k 10 120 python3 tools/kbench.py c2n10 demo10 c3n10 --rounds 6 2>&1 | grep -v amdgpu.ids | cut -c1-110
done | tee $OUT/sweep.txt
