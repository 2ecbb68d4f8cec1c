#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05h
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
for W in 0 6 8; do
  if [ $W = 0 ]; then unset PTRACE_LIB; echo "== shipped (91 VGPRs)"; else export PTRACE_LIB=$ROOT/build_variants/libptrace_t4w$W.so; echo "== waves_per_eu($W, 8)"; fi
  timeout -k 10 100 python3 tools/kbench.py c2 c2onoff --rounds 30 2>/dev/null | cut -c1-100
  timeout -k 10 100 python3 bench.py --no-extras --no-cpu-baseline --no-in-flight 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench 200-step ms_per_step', round(d['ms_per_step']*1e3,3), 'us; avg kernel', round(d['roofline']['avg_kernel_ms']*1e3,3), 'parity', d['parity_check']['bit_identical'])"
done; done 2>&1 | tee $OUT/tile4_waves.txt
