#!/bin/bash
# what a timed region costs beyond its frames: signal waits by interrupt (default) or by polling (HSA_ENABLE_INTERRUPT=0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05i
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
for mode in default poll; do
  if [ $mode = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-in-flight 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', 'K=20 ms_per_step', round(d['ms_per_step']*1e3,3), 'us; at 10K', round(d['same_loop_at_10x_steps']['ms_per_step']*1e3,3), 'fixed cost us', round(d['same_loop_at_10x_steps']['fixed_cost_per_timed_region_us'],1), 'parity', d['parity_check']['bit_identical'])"
done; done 2>&1 | tee $OUT/sync_mode.txt
unset HSA_ENABLE_INTERRUPT
PTRACE_LIB=$ROOT/build_variants/libptrace_dbg.so DBG_RENDERER=flat DBG_PLANE=1 DBG_S=0 timeout -k 10 100 python3 tools/dbgtime.py 2>&1 | tail -12 | tee $OUT/dbgtime_tile4.txt
