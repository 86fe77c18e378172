#!/bin/bash
# How often does tests/test_gpu_pipeline.py::test_graph_replay_matches_eager[megapose-2] fail, and how many of N two-lane graph
# replays differ from the first (tools/probes/two_lane_repro.py), for each library given (A/B on ONE box)?
#   tools/probes/graph_flake.sh <lib.so> [<lib.so> ...]
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  fails=0
  for i in 1 2 3 4 5 6; do
    HAPPYPOSE_AMD_LIB=$lib python3 -m pytest "tests/test_gpu_pipeline.py::test_graph_replay_matches_eager" -q -m gpu -k "megapose-2" > /tmp/flake.log 2>&1 || fails=$((fails+1))
  done
  echo "$lib: test failed $fails of 6"
  for e in HP_X=1 HP_RASTER_NO_CULL=1; do
    echo "  $e $(env $e HP_PROBE_PIXELS=0 HAPPYPOSE_AMD_LIB=$lib python3 tools/probes/two_lane_repro.py 300 2 1 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['first_difference_tally'])")"
  done
done
