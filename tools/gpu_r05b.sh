#!/bin/bash
# round 5, second GPU call: the plan + enqueue refactor against the GPU suite and the kernel times of every configuration;
# select-on-VCC issue cost; the profiling pipeline (code hash + disassembly bounds) on the headline kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05b
mkdir -p $OUT
cd $ROOT
timeout -k 10 120 tools/micro/issue2 > $OUT/issue2.txt 2>&1; echo "issue2 rc $?"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -rs > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/gpu_tests.log
timeout -k 10 300 python3 tools/kbench.py c2 c2onoff c2s2 c3 c3:sample c3n10 c2n10 demo10 c4 c4:sample c4rank c4rank:sample c5 c5pt pl pl5 c2ortho c3ortho --rounds 8 > $OUT/kbench.txt 2>&1; cat $OUT/kbench.txt | cut -c1-110
timeout -k 10 500 bash tools/prof_bench.sh r05b c2-only > $OUT/prof.log 2>&1; echo "prof rc $?"; tail -5 $OUT/prof.log | cut -c1-300
cp gpurun_out/prof_r05b/pmc_c2.json profiles/pmc_c2.json
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench rc $?"
