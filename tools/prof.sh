#!/bin/bash
# Profile the kernel micro-benchmark on the GPU box: kernel trace + two PMC passes (separate runs).
# usage: tools/prof.sh <tag> <kbench args...>
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/kbench.py "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 tools/kbench.py "$@" --rounds 2 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc2 -- python3 tools/kbench.py "$@" --rounds 2 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 tools/kbench.py "$@" --rounds 2 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 tools/kbench.py "$@" --rounds 2 > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
