#!/bin/bash
# round 5, first GPU call: unmeasured opcode prices, the driver's bench command with the pre-roll, the GPU suite, PC-sampling support
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05a
mkdir -p $OUT
cd $ROOT
timeout -k 10 120 tools/micro/issue2 > $OUT/issue2.txt 2>&1; echo "issue2 rc $?"
timeout -k 10 60 rocprofv3-avail list --pc-sampling > $OUT/pcsamp_avail.txt 2>&1 || timeout -k 10 60 rocprofv3-avail -h > $OUT/pcsamp_avail_help.txt 2>&1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 rc $?"
timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline > $OUT/bench200.json 2> $OUT/bench200.err; echo "bench200 rc $?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pre-roll-ms 0 --min-timed-ms 0 > $OUT/bench20_nopreroll.json 2> $OUT/bench20_nopreroll.err; echo "bench20 no pre-roll rc $?"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -rs > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/gpu_tests.log
timeout -k 10 200 python3 tools/kbench.py c2 c3n10 c2n10 demo10 c3 c3:sample c4:sample --rounds 8 > $OUT/kbench.txt 2>&1; tail -8 $OUT/kbench.txt
