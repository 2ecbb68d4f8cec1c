#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05e
mkdir -p $OUT
cd $ROOT
export PTRACE_LIB=$ROOT/build_variants/libptrace_dbg.so
DBG_LANES=1 timeout -k 10 120 python3 tools/dbgtree.py c3n10 2>&1 | tail -12 | tee $OUT/dbgtree.txt
PTRACE_QCHOICE=2 DBG_LANES=1 DBG_PLANE=1 DBG_S=1 DBG_N=10 timeout -k 10 120 python3 tools/dbgtime.py 2>&1 | tail -13 | tee $OUT/dbgtime_flagged.txt
