#!/bin/bash
# per-library flake statistics of eager two-lane steps with the renders kept (first differing field tallied): P processes x R runs
cd $GRAFT_REPO_ROOT
P=${P:-5}; R=${R:-200}
for lib in "$@"; do
  bad=0; tot=0; keys=""
  for i in $(seq $P); do
    out=$(HAPPYPOSE_AMD_LIB=$lib HP_PROBE_PIXELS=${PIX:-1} python3 tools/probes/two_lane_repro.py $R 2 ${GRAPHS:-0} 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); t=d['first_difference_tally']; print(sum(t.values()), ','.join(sorted(t)))")
    n=${out%% *}; tot=$((tot+n)); [ "$n" -gt 0 ] && bad=$((bad+1)); keys="$keys ${out#* }"
  done
  echo "$lib: $bad of $P processes flaky, $tot differing runs of $((P*R)); first differing fields:$keys"
done
