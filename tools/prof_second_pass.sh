#!/bin/bash
# Where the path tracer's second pass waits: PMC passes over one kbench configuration (each pass its own run).
# usage: tools/prof_second_pass.sh <tag> <config>      e.g.  r02g c3:sample      (run on the GPU box, from the repository root)
set -e
TAG=$1
CFG=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sp_${TAG}_${CFG//:/_}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
KB="python3 $ROOT/tools/kbench.py $CFG --rounds 3"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- $KB > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -- $KB > $OUT/b.log 2>&1
# (no TA_* pass: a run with TA_TA_BUSY_sum / TA_*_WAVEFRONTS_sum aborted inside hipMalloc on this pool and then hung)
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_VALU --output-format csv -d $OUT/d -- $KB > $OUT/d.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $OUT/a $OUT/b $OUT/d --kernel "pt_path_regions_kernel" --json $OUT/summary.json --source "tools/prof_second_pass.sh $TAG $CFG" > /dev/null
cat $OUT/summary.json
