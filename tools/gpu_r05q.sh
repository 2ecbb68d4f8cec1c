#!/bin/bash
# per-kernel times of the hand-over frames (rocprofv3 kernel trace)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "demo10 600" "demo10 800" "c2n10 400"; do
  set -- $cfg
  export PTRACE_Q_BUDGET=$2
  rm -rf /tmp/prof_q
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -- python3 $ROOT/tools/kbench.py $1 --rounds 6 > /dev/null 2>&1
  echo "== $1 budget $2"
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/prof_q/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>0.5: print(r["Name"][:70], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1))
PY
done 2>&1 | tee $OUT/kernel_split.txt
