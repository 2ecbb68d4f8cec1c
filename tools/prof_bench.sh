#!/bin/bash
# Profile bench.py itself on the GPU box: kernel trace of the DEFAULT command (the one the driver runs), then PMC passes (each
# in its own run, nothing combined with a trace domain), condensed into the JSON files bench.py reads.  Every JSON records the
# sha256 of the device code it was collected on (tools/pmc_summary.py --lib): bench.py prices nothing from a file of another build.
# usage: tools/prof_bench.sh <tag> [trace-only | c2-only]   (run from the repository root on the GPU box)
set -e
TAG=$1
MODE=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_SMEM"
P3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"
# the kernel trace runs bench.py's DEFAULT command: the launches of the headline kernel in it are the pre-roll and the timed loops
# (back to back), a few per cent of them the in-flight and dome-off side rows
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv | cut -d, -f1-4
if [ "$MODE" = "trace-only" ]; then exit 0; fi
# PMC on the headline kernel: the same loops, a few hundred dispatches (counters are per dispatch; medians)
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-in-flight --headline-only --pre-roll-ms 3 --min-timed-ms 3"
rocprofv3 --pmc $P1 --output-format csv -d $OUT/pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --pmc $P2 --output-format csv -d $OUT/pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --pmc $P3 --output-format csv -d $OUT/pmc2b -- $BENCH > $OUT/pmc2b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- $BENCH > $OUT/pmc4.log 2>&1
SRC="rocprofv3 --pmc (five separate passes: SQ issue counters, fp64 / integer instruction classes, fp32 / conversion classes, FETCH_SIZE, WRITE_SIZE) on '$BENCH'; medians over the dispatches of the kernel; tools/prof_bench.sh $TAG"
( cd $ROOT && python3 tools/pmc_summary.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc2b $OUT/pmc3 $OUT/pmc4 --kernel "pt_tile4_kernel<1, true>" --grid 235520 --json $OUT/pmc_c2.json --source "$SRC" > /dev/null )
( cd $ROOT && python3 tools/isa_mix.py "pt_tile4_kernel<1, true>" --pmc $OUT/pmc_c2.json --update > $OUT/isa_mix_c2.txt 2>&1 ) || true
if [ "$MODE" = "c2-only" ]; then cat $OUT/pmc_c2.json; exit 0; fi
# the other kernels, each through tools/kbench.py (one configuration, several rounds in one process): three PMC passes
#   name : kbench configuration : kernel substring : json : what it is
while IFS='|' read -r name cfg kern json what; do
  [ -z "$name" ] && continue
  KB="python3 $ROOT/tools/kbench.py $cfg --rounds 4"
  rocprofv3 --pmc $P1 --output-format csv -d $OUT/pmc_$name -- $KB > $OUT/pmc_$name.log 2>&1
  rocprofv3 --pmc $P2 --output-format csv -d $OUT/pmc_${name}_2 -- $KB > $OUT/pmc_${name}_2.log 2>&1
  rocprofv3 --pmc $P3 --output-format csv -d $OUT/pmc_${name}_3 -- $KB > $OUT/pmc_${name}_3.log 2>&1
  ( cd $ROOT && python3 tools/pmc_summary.py $OUT/pmc_$name $OUT/pmc_${name}_2 $OUT/pmc_${name}_3 --kernel "$kern" --json $OUT/$json \
      --source "rocprofv3 --pmc (three passes) on '$KB' ($what); medians over the launches; tools/prof_bench.sh $TAG" > /dev/null ) || echo "no summary for $name"
  # the kernel's disassembly priced per instruction, block counts bounded by these counters (tools/isa_mix.py)
  ( cd $ROOT && K=$(python3 -c "import json,sys; print(json.load(open('$OUT/$json'))['kernel'].replace('void ','').split('(')[0])") && \
      timeout -k 5 300 python3 tools/isa_mix.py "$K" --pmc $OUT/$json --update > $OUT/isa_mix_$name.txt 2>&1 ) || echo "no disassembly bounds for $name"
done <<'CFG'
c3|c3|pt_path_regions_kernel|pmc_c3_second_pass.json|C3, PT_PCG_PIXEL
c3s|c3:sample|pt_path_regions_kernel|pmc_c3_second_pass_sample.json|C3, PT_PCG_SAMPLE
c4s|c4:sample|pt_path_regions_kernel|pmc_c4_second_pass_sample.json|C4: 3840x2160, 256 spheres, D = 5, spp 64, PT_PCG_SAMPLE
tree|c3n10:tree|pt_path_tree_kernel|pmc_c3n10_tree.json|C3 scene, PathTracer N = 10, D = 3, S = 1: the CLI's defaults, the tree kernel rendering the frame alone (qchoice 0; at its 29 k flagged pixels the device lets the one-queue kernel start the frame since round 5)
c2n10|c2n10|pt_path_flagged_kernel|pmc_c2n10_flagged.json|C2 scene with its ground plane, PathTracer N = 10, D = 3, S = 1: 490 k flagged pixels, the device picks the one-queue kernel
pl|pl|pt_tile_kernel<3, 3, false, false|pmc_pointlight_tile.json|C2 scene + two point lights, PointLightRenderer: primary + shadow rays
c3ortho|c3ortho|pt_path_regions_kernel|pmc_c3ortho_second_pass.json|C3 scene through an orthogonal camera: first pass with beams, second pass by regions
CFG
# C5 (10 000 spheres, Flat): HBM bytes of its two kernels (BASELINE.json configs[4] asks for rocprof HBM GB/s)
KB="python3 $ROOT/tools/kbench.py c5 --rounds 4"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_c5f -- $KB > $OUT/pmc_c5f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_c5w -- $KB > $OUT/pmc_c5w.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_c5f $OUT/pmc_c5w --kernel "pt_tile_kernel<1, 4, true, false" --json $OUT/pmc_c5_tile.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) on '$KB' (C5: 1280x720, 10 000 spheres, Flat); medians over the launches; tools/prof_bench.sh $TAG" > /dev/null
python3 tools/pmc_summary.py $OUT/pmc_c5f $OUT/pmc_c5w --kernel "pt_cell_kernel" --json $OUT/pmc_c5_cell.json --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) on '$KB'; medians over the launches; tools/prof_bench.sh $TAG" > /dev/null
cat $OUT/pmc_c2.json
