#!/bin/bash
# round 5, fourth GPU call: every v_cndmask_b32 in its VOP3 encoding (build_variants/libptrace_e64.so) against the shipped build;
# section cycles of the one-queue and the tree kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05d
mkdir -p $OUT
cd $ROOT
CFG="c2 c2onoff c2s2 c3 c3:sample c3n10 c2n10 demo10 c4 c4:sample c4rank:sample c5 c5pt pl pl5 c2ortho c3ortho"
for rep in 1 2; do
echo "== shipped build ($rep)"; timeout -k 10 300 python3 tools/kbench.py $CFG --rounds 10 2>/dev/null | cut -c1-100
echo "== VOP3 cndmask ($rep)"; PTRACE_LIB=$ROOT/build_variants/libptrace_e64.so timeout -k 10 300 python3 tools/kbench.py $CFG --rounds 10 2>/dev/null | cut -c1-100
done > $OUT/kbench_e64.txt 2>&1
cat $OUT/kbench_e64.txt
PTRACE_LIB=$ROOT/build_variants/libptrace_e64.so timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests_e64.log 2>&1; echo "pytest e64 rc $?"; tail -2 $OUT/gpu_tests_e64.log
export PTRACE_LIB=$ROOT/build_variants/libptrace_dbg.so
echo "== tree kernel sections (c3n10)"; timeout -k 10 120 python3 tools/dbgtree.py c3n10 2>&1 | tail -12 | tee $OUT/dbgtree.txt
echo "== one-queue kernel sections (C2 + plane, N = 10, forced)"; PTRACE_QCHOICE=2 DBG_PLANE=1 DBG_S=1 DBG_N=10 timeout -k 10 120 python3 tools/dbgtime.py 2>&1 | tail -12 | tee $OUT/dbgtime_flagged.txt
