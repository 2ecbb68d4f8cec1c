#!/bin/bash
# PMC passes on the path tracer's second pass of C4 under PT_PCG_SAMPLE (pt_path_regions_kernel<true, false, 2>): instruction
# counts by class and duration per launch -> gpurun_out/prof_<tag>/pmc_c4_second_pass_sample.json (copy it to profiles/).
# usage: tools/prof_c4.sh <tag>     (repository root, on the GPU box)
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
KB="python3 $ROOT/tools/kbench.py c4:sample --rounds 4"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_c4s -- $KB > $OUT/pmc_c4s.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_c4s_2 -- $KB > $OUT/pmc_c4s_2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_c4s_3 -- $KB > $OUT/pmc_c4s_3.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_c4s $OUT/pmc_c4s_2 $OUT/pmc_c4s_3 --kernel "pt_path_regions_kernel" --json $OUT/pmc_c4_second_pass_sample.json --source "rocprofv3 --pmc (three passes) on 'python3 tools/kbench.py c4:sample --rounds 4' (C4: 3840x2160, 256 spheres, D = 5, spp 64, PT_PCG_SAMPLE); medians over the launches; tools/prof_c4.sh $TAG" > /dev/null
cat $OUT/pmc_c4_second_pass_sample.json
