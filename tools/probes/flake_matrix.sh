#!/bin/bash
# Per-process flake statistics of two-lane graph replays under environment switches: each setting gets P fresh processes of
# R runs (the condition that makes a process flaky is established at start-up: rates are 0 or ~5 % per process).
#   tools/probes/flake_matrix.sh "HP_X=1" "HP_RASTER_NO_CULL=1" ... (switches of csrc/debug.h)
cd $GRAFT_REPO_ROOT
P=${P:-5}; R=${R:-150}
for e in "$@"; do
  bad=0; tot=0
  for i in $(seq $P); do
    n=$(env $e HP_PROBE_PIXELS=0 python3 tools/probes/two_lane_repro.py $R 2 1 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sum(d['first_difference_tally'].values()))")
    tot=$((tot+n)); [ "$n" -gt 0 ] && bad=$((bad+1))
  done
  echo "$e: $bad of $P processes flaky, $tot differing runs of $((P*R))"
done
