#!/bin/bash
# round-3 probe: which PMC counter tracks VALU issue cycles; where pt_tile_kernel<FLAT>'s time goes; grid-size sweep
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > $OUT/counters.txt 2>&1 || true
hipcc --offload-arch=gfx950 -O2 -o /tmp/issue $ROOT/tools/micro/issue.hip 2>/dev/null
export ISSUE_WS=4
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_issue -- /tmp/issue > $OUT/pmc_issue.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_issue2 -- /tmp/issue > $OUT/pmc_issue2.log 2>&1 || true
unset ISSUE_WS
cd $ROOT
PTRACE_LIB=$ROOT/build_variants/libptrace_dbg.so python3 tools/dbgtime_tile.py > $OUT/dbgtime_tile.txt 2>&1 || true
for g in 4 5 6 8 10 12 15; do echo "PTRACE_TILE_WG_PER_CU=$g"; PTRACE_TILE_WG_PER_CU=$g python3 tools/kbench.py c2 c2onoff c2s2 --rounds 30; done > $OUT/grid_sweep.txt 2>&1
