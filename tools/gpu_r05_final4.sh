#!/bin/bash
# round 5, final evidence after the last kernel commit (the hand-over of heavy pixels included): suite, profiles (trace + PMC + disassembly bounds), bench lines, kernel times, sections
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_final
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -rs -s > $OUT/gpu_tests_final.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/gpu_tests_final.log | cut -c1-200
timeout -k 10 900 bash tools/prof_bench.sh r05 > $OUT/prof.log 2>&1; echo "prof rc $?"; tail -3 $OUT/prof.log | cut -c1-200
cp gpurun_out/prof_r05/pmc_*.json profiles/ 2>/dev/null
cp gpurun_out/prof_r05/kernel_stats.csv $OUT/bench_kernel_stats.csv
timeout -k 10 600 python3 bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "bench rc $?"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "bench (driver's command) rc $?"
timeout -k 10 400 python3 tools/kbench.py c2 c2onoff c2s2 c3 c3:sample c3n10 c2n10 demo10 c4 c4:sample c4rank c4rank:sample c5 c5pt pl pl5 c2ortho c3ortho --rounds 10 > $OUT/kbench_final.txt 2>&1; cat $OUT/kbench_final.txt | cut -c1-105
export PTRACE_LIB=$ROOT/build_variants/libptrace_dbg.so
timeout -k 10 120 python3 tools/dbgtile4_waves.py > $OUT/tile4_wave_cycles.txt 2>&1
DBG_LANES=1 timeout -k 10 120 python3 tools/dbgtree.py c3n10 2>&1 | tail -11 | tee $OUT/dbgtree.txt
PTRACE_QCHOICE=2 PTRACE_Q_BUDGET=0 PTRACE_Q_TAIL_BUDGET=0 PTRACE_Q_FEW_LANES=0 DBG_LANES=1 DBG_PLANE=1 DBG_S=1 DBG_N=10 timeout -k 10 120 python3 tools/dbgtime.py 2>&1 | tail -13 | tee $OUT/dbgtime_flagged.txt
unset PTRACE_LIB
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $OUT/smoke.log
