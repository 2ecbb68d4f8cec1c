#!/bin/bash
# round 5, third GPU call: the split frame stack of the one-queue kernel (correctness, then home 0 / 1 / 2 timed); PMC probe of issue2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05c
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
# the one-queue kernel forced on every num_of_rays > 1 frame, split stack (default) -- the whole GPU suite
PTRACE_QCHOICE=2 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests_queue_forced_split.log 2>&1; echo "pytest (queue forced, split) rc $?"; tail -2 $OUT/gpu_tests_queue_forced_split.log
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/gpu_tests.log
for home in 1 2 0; do
  echo "== PTRACE_Q_FRAMES_HOME=$home (queue forced)"
  PTRACE_QCHOICE=2 PTRACE_Q_FRAMES_HOME=$home timeout -k 10 200 python3 tools/kbench.py c2n10 demo10 c3n10 --rounds 6 2>/dev/null | cut -c1-100
done > $OUT/q_home.txt 2>&1
cat $OUT/q_home.txt
echo "== the device's choice"; timeout -k 10 200 python3 tools/kbench.py c2n10 demo10 c3n10 --rounds 6 2>/dev/null | cut -c1-100 | tee $OUT/q_choice.txt
timeout -k 10 300 python3 tools/tree_vs_queue.py > $OUT/tree_vs_queue.txt 2>&1; tail -40 $OUT/tree_vs_queue.txt
cd /tmp
ISSUE_WS=4 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_issue2 -- $ROOT/tools/micro/issue2 > $OUT/pmc_issue2.log 2>&1; echo "pmc issue2 rc $?"
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_issue2 > $OUT/pmc_issue2_rows.txt 2>&1; grep -c . $OUT/pmc_issue2_rows.txt
