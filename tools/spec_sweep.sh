#!/bin/bash
# PT_PCG_PIXEL speculation: kernel times (production build) and speculation statistics (debug build) over the
# window-speculation switches.  usage (GPU box): bash tools/spec_sweep.sh > gpurun_out/spec_sweep.txt
cd $GRAFT_REPO_ROOT
for cfg in "65 13 0" "4 13 0" "8 13 0" "4 10 0" "4 16 0" "8 13 4" "8 13 2" "4 13 4"; do
  set -- $cfg
  export PTRACE_SPEC_WIN_LANES=$1 PTRACE_SPEC_WIN_COVER=$2 PTRACE_UNIT_MIN_ROUNDS=$3
  echo "=== WIN_LANES=$1 COVER=$2 MIN_ROUNDS=$3"
  python3 tools/kbench.py c3 c4rank c4 c3:sample --rounds 6 2>/dev/null | cut -c1-110
  PTRACE_LIB=build_variants/libptrace_dbg.so python3 tools/dbgspec.py c3 c4rank 2>/dev/null | grep -v "cycles per wave"
  PTRACE_LIB=build_variants/libptrace_dbg.so python3 tools/dbgunits.py c3 c4rank 2>/dev/null | grep "unit duration" | cut -c1-200
done
