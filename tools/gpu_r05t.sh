#!/bin/bash
# scatter once behind the unwind loop + sincos: the equality probe, kernel times (hand-over off), then parity
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05t
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_probes.py -m gpu -x -q -k "sincos or scatter" > $OUT/probe.log 2>&1; echo "probe rc $?"; tail -3 $OUT/probe.log | cut -c1-200
export PTRACE_Q_BUDGET=0 PTRACE_Q_TAIL_BUDGET=0 PTRACE_Q_FEW_LANES=0
timeout -k 10 300 python3 tools/kbench.py c3 c3:sample c3n10 c2n10 demo10 c4 c4:sample c4rank c4rank:sample c5pt c3ortho --rounds 8 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee $OUT/kbench.txt
unset PTRACE_Q_BUDGET PTRACE_Q_TAIL_BUDGET PTRACE_Q_FEW_LANES
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m "gpu and not slow" -x -q > $OUT/tests.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/tests.log | cut -c1-200
