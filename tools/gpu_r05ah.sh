#!/bin/bash
# pt_tile4_kernel: planes first, spheres dismissed without their roots where certain: parity, then times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05ah
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m "gpu and not slow" -x -q > $OUT/tests.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/tests.log | cut -c1-200
timeout -k 10 200 python3 tools/kbench.py c2 c2onoff c2ortho c2s2 --rounds 20 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee $OUT/kbench.txt
timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --no-in-flight 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=200 us per frame', round(d['ms_per_step']*1e3,3), 'value', round(d['value']), 'parity', d['parity_check']['bit_identical'], 'frac', d['roofline']['frac'])" | tee -a $OUT/kbench.txt
PT_FUZZ_SEEDS=200 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k random_scenes > $OUT/fuzz200.log 2>&1; echo "fuzz rc $?"; tail -2 $OUT/fuzz200.log
