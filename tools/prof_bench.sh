#!/bin/bash
# Profile bench.py itself on the GPU box: kernel trace + PMC passes (each in its own run, nothing combined
# with a trace domain), then condense them into the JSON files bench.py reads.
# usage: tools/prof_bench.sh <tag> [trace-only]   (run from the repository root on the GPU box)
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
# the kernel trace runs bench.py's DEFAULT command (the one the driver runs): most launches of the headline kernel
# in it are the timed ones, so its average is comparable with roofline.avg_kernel_ms of the line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err
if [ "$2" = "trace-only" ]; then
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/kernel_stats.csv
  head -12 $OUT/kernel_stats.csv | cut -d, -f1-4
  exit 0
fi
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc2b -- $BENCH > $OUT/pmc2b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- $BENCH > $OUT/pmc4.log 2>&1
# the path tracer's second pass on C3 alone (both PCG modes): instruction counts and duration per launch
KB="python3 $ROOT/tools/kbench.py c3 --rounds 4"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_c3 -- $KB > $OUT/pmc_c3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_c3_2 -- $KB > $OUT/pmc_c3_2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_c3_3 -- $KB > $OUT/pmc_c3_3.log 2>&1
KB="python3 $ROOT/tools/kbench.py c3:sample --rounds 4"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_c3s -- $KB > $OUT/pmc_c3s.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_c3s_2 -- $KB > $OUT/pmc_c3s_2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_c3s_3 -- $KB > $OUT/pmc_c3s_3.log 2>&1
# the CLI's default path tracer (N = 10, D = 3, one sample per pixel) on the C3 scene: pt_path_tree_kernel
KB="python3 $ROOT/tools/kbench.py c3n10 --rounds 4"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_tree -- $KB > $OUT/pmc_tree.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_tree2 -- $KB > $OUT/pmc_tree2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_tree3 -- $KB > $OUT/pmc_tree3.log 2>&1
# primary + shadow rays (C2 scene + two point lights): pt_tile_kernel<POINTLIGHT>; orthogonal path tracing: pt_path_kernel;
# a frame FULL of flagged pixels at the CLI's N = 10, D = 3 (C2 scene with its ground plane): pt_path_flagged_kernel
for cfg in pl c3ortho c2n10; do
  KB="python3 $ROOT/tools/kbench.py $cfg --rounds 4"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_$cfg -- $KB > $OUT/pmc_$cfg.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_${cfg}_2 -- $KB > $OUT/pmc_${cfg}_2.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_${cfg}_3 -- $KB > $OUT/pmc_${cfg}_3.log 2>&1
done
# C5 (10 000 spheres, Flat): HBM bytes of its two kernels (BASELINE.json configs[4] asks for rocprof HBM GB/s)
KB="python3 $ROOT/tools/kbench.py c5 --rounds 4"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_c5f -- $KB > $OUT/pmc_c5f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_c5w -- $KB > $OUT/pmc_c5w.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_pl $OUT/pmc_pl_2 $OUT/pmc_pl_3 --kernel "pt_tile_kernel<3, 3, false, false" --json $OUT/pmc_pointlight_tile.json --source "rocprofv3 --pmc (three passes) on 'python3 tools/kbench.py pl --rounds 4' (C2 scene + two point lights, PointLightRenderer: primary + shadow rays); medians over the launches; tools/prof_bench.sh $TAG" > /dev/null || true
python3 tools/pmc_summary.py $OUT/pmc_c3ortho $OUT/pmc_c3ortho_2 $OUT/pmc_c3ortho_3 --kernel "pt_path_regions_kernel" --json $OUT/pmc_c3ortho_second_pass.json --source "rocprofv3 --pmc (three passes) on 'python3 tools/kbench.py c3ortho --rounds 4' (C3 scene through an orthogonal camera: first pass with beams, second pass by regions); medians over the launches; tools/prof_bench.sh $TAG" > /dev/null || true
python3 tools/pmc_summary.py $OUT/pmc_c2n10 $OUT/pmc_c2n10_2 $OUT/pmc_c2n10_3 --kernel "pt_path_flagged_kernel" --json $OUT/pmc_c2n10_flagged.json --source "rocprofv3 --pmc (three passes) on 'python3 tools/kbench.py c2n10 --rounds 4' (C2 scene with its ground plane, PathTracer N = 10, D = 3, S = 1: 490 k flagged pixels, the device picks the one-queue kernel); medians over the launches; tools/prof_bench.sh $TAG" > /dev/null || true
python3 tools/pmc_summary.py $OUT/pmc_c5f $OUT/pmc_c5w --kernel "pt_tile_kernel<1, 4, true, false" --json $OUT/pmc_c5_tile.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) on 'python3 tools/kbench.py c5 --rounds 4' (C5: 1280x720, 10 000 spheres, Flat); medians over the launches; tools/prof_bench.sh $TAG" > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_c5f $OUT/pmc_c5w --kernel "pt_cell_kernel" --json $OUT/pmc_c5_cell.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) on 'python3 tools/kbench.py c5 --rounds 4'; medians over the launches; tools/prof_bench.sh $TAG" > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_tree $OUT/pmc_tree2 $OUT/pmc_tree3 --kernel "pt_path_tree_kernel" --json $OUT/pmc_c3n10_tree.json --source "rocprofv3 --pmc (three passes) on 'python3 tools/kbench.py c3n10 --rounds 4' (C3 scene, PathTracer N = 10, D = 3, S = 1: the CLI's defaults); medians over the launches; tools/prof_bench.sh $TAG" > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_c3 $OUT/pmc_c3_2 $OUT/pmc_c3_3 --kernel "pt_path_regions_kernel" --json $OUT/pmc_c3_second_pass.json --source "rocprofv3 --pmc on 'python3 tools/kbench.py c3 --rounds 4' (C3, PT_PCG_PIXEL); medians over the launches; tools/prof_bench.sh $TAG" > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_c3s $OUT/pmc_c3s_2 $OUT/pmc_c3s_3 --kernel "pt_path_regions_kernel" --json $OUT/pmc_c3_second_pass_sample.json --source "rocprofv3 --pmc on 'python3 tools/kbench.py c3:sample --rounds 4' (C3, PT_PCG_SAMPLE); medians over the launches; tools/prof_bench.sh $TAG" > /dev/null
SRC="rocprofv3 --pmc (five separate passes: SQ issue counters, fp64 / integer instruction classes, fp32 / conversion classes, FETCH_SIZE, WRITE_SIZE) on 'python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline'; medians over the dispatches of the kernel; tools/prof_bench.sh $TAG"
python3 tools/pmc_summary.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc2b $OUT/pmc3 $OUT/pmc4 --kernel "pt_tile4_kernel<1, true, 4>" --grid 235520 --json $OUT/pmc_c2.json --source "$SRC" > /dev/null
python3 tools/pmc_summary.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc2b $OUT/pmc3 $OUT/pmc4 --kernel "pt_path_regions_kernel" --grid 131072 --json $OUT/pmc_path_second_pass.json --source "$SRC (all second-pass launches of the run: C3, C4, shares)" > /dev/null || true
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -d, -f1-4
cat $OUT/pmc_c2.json
