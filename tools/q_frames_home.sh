#!/bin/bash
# The one-queue second pass with its frame stack in LDS (one workgroup per CU at D = 3) against the
# same stack in HBM (two): kernel ms of dense frames, the queue kernel forced.  usage (GPU box): bash tools/q_frames_home.sh
cd $GRAFT_REPO_ROOT
for home in 1 0; do
  echo "== PTRACE_Q_LDS_FRAMES=$home"
  for c in "plane 1280 720 10 3 1" "demo 1280 960 10 3 1" "plane 1280 720 4 3 1" "plane 1280 720 2 3 1" "plane 1280 720 10 2 1" "plane 1920 1080 10 3 1" "demo 1280 960 3 3 2"; do
    PTRACE_QCHOICE=2 PTRACE_Q_LDS_FRAMES=$home python3 tools/tree_vs_queue.py --one $c 2>/dev/null | awk -v c="$c" '{print c, "->", $1, "ms"}'
  done
done
