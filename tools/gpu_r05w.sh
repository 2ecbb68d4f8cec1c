#!/bin/bash
# defaults (budget 400, tail 50, few lanes 16): kernel times, then the tree / queue table for the crossover
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05w
mkdir -p $OUT
cd $ROOT
timeout -k 10 200 python3 tools/kbench.py c3n10 c2n10 demo10 c3 c3:sample --rounds 10 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee $OUT/kbench.txt
timeout -k 10 900 python3 tools/tree_vs_queue.py 2>&1 | grep -v amdgpu.ids | tee $OUT/tree_vs_queue.txt
