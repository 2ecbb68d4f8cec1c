#!/bin/bash
# The ONE parametrised runner for a GPU box (gpurun -- 'bash tools/gpu_run.sh STEP [STEP ...]'): each step writes its log under
# gpurun_out/<tag>/ and the steps are chained with && semantics (set -e: a failed or killed GPU step starts no further one).
#   tests[:EXPR]      python -m pytest tests -m gpu -x -q [-k EXPR]
#   bench             the driver's command: python bench.py (N = 1), line -> bench.json, detail -> bench_detail.json
#   bench20           python bench.py --steps 20 --warmup 5 (the driver's K)
#   prof[:MODE]       tools/prof_bench.sh <tag> [trace-only | c2-only]: kernel trace + PMC passes -> pmc_*.json
#   gloo2[:FAIL]      PT_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 4 --warmup 2 (two ranks share the card);
#                     FAIL = "timed sparse:raise" | "timed sparse:hang" rehearses the fallback paths (PT_BENCH_FAIL)
#   kbench:CFG        python tools/kbench.py CFG --rounds 4
#   ab:LIBS:CFGS      A/B of library builds: for every name in LIBS (commas; `cur` = pytracer_amd/libptrace.so, anything else
#                     build_variants/libptrace_<name>.so, tools/build_variant.sh) tools/kbench.py CFGS (commas) --rounds 8, the whole
#                     list twice, alternating -> ab_<LIBS>.txt
#   py:SCRIPT[:ARGS]  python SCRIPT ARGS (ARGS separated by commas)
# TAG (environment, default r06) names the output directory.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r06}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for step in "$@"; do
  name=${step%%:*}
  arg=""
  [ "$name" != "$step" ] && arg=${step#*:}
  echo "== $step"
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q -k "$arg" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
      else timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }; fi
      tail -3 $OUT/tests.log ;;
    bench)
      timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
      cp bench_detail.json $OUT/bench_detail.json; tail -c 4200 $OUT/bench.json ;;
    bench20)
      timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err || { tail -5 $OUT/bench20.err; exit 1; }
      cp bench_detail.json $OUT/bench20_detail.json; tail -c 4200 $OUT/bench20.json ;;
    prof)
      timeout -k 10 1100 bash tools/prof_bench.sh $TAG $arg > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
      tail -5 $OUT/prof.log ;;
    gloo2)
      log=$OUT/gloo2${arg:+_$(echo "$arg" | tr ' :' '__')}.log
      PT_BENCH_FAIL="$arg" PT_BENCH_PHASE_S=${PT_BENCH_PHASE_S:-90} PT_DIST_BACKEND=gloo timeout -k 10 900 python3 bench.py --gpus 2 --steps 4 --warmup 2 > $log 2>&1 || { echo "exit $?"; grep -v "^\[bench detail\]" $log | tail -8 | cut -c1-600; exit 1; }
      cp bench_detail.json ${log%.log}_detail.json; grep "^\[bench " $log | cut -c1-200; tail -1 $log | cut -c1-4200 ;;
    kbench)
      timeout -k 10 600 python3 tools/kbench.py $arg --rounds 4 > $OUT/kbench_$(echo "$arg" | tr ' :' '__').txt 2>&1; tail -12 $OUT/kbench_$(echo "$arg" | tr ' :' '__').txt ;;
    ab)
      libs=$(echo "${arg%%:*}" | tr ',' ' '); cfgs=$(echo "${arg#*:}" | tr ',' ' ')
      log=$OUT/ab_$(echo "${arg%%:*}" | tr ',' '_').txt
      : > $log
      for pass in 1 2; do for l in $libs; do
        lib=pytracer_amd/libptrace.so; [ "$l" != "cur" ] && lib=build_variants/libptrace_$l.so
        echo "== $l (pass $pass)" >> $log
        PTRACE_LIB=$lib timeout -k 10 600 python3 tools/kbench.py $cfgs --rounds 8 2>&1 | cut -c1-120 >> $log
      done; done
      cat $log ;;
    py)
      script=${arg%%:*}; rest=""; [ "$script" != "$arg" ] && rest=$(echo "${arg#*:}" | tr ',' ' ')
      timeout -k 10 900 python3 $script $rest > $OUT/$(basename $script .py).txt 2>&1 || { tail -20 $OUT/$(basename $script .py).txt; exit 1; }
      tail -40 $OUT/$(basename $script .py).txt ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
