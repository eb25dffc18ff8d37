#!/bin/bash
# The GPU tier four ways, every round: defaults, the latency variants forced wherever a graph allows them (lat=2), forbidden (lat=0),
# and the search launches with their visited sets kept (novis=0).  Each way must be green: the switches pick between traversal forms that
# are all held to the same oracle.  Summary -> gpurun_out/gpu_tiers.log (copied to profiles/r<N>_gpu_tiers.log).
#   usage: tools/run_tiers.sh [pytest selection ...]      (default: the whole GPU tier without the full-size builds for the forced ways)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; L=$O/gpu_tiers.log; : > $L
SEL=${@:-tests}
way() { # name, diag, selection
  echo "== $1 (HNSW_MI355X_DIAG='$2')" >> $L
  HNSW_MI355X_DIAG="$2" timeout -k 10 ${TIER_TIMEOUT:-900} python3 -m pytest $3 -m gpu -q -x -p no:cacheprovider > $O/gpu_tier_$1.log 2>&1
  echo "rc=$? $(tail -1 $O/gpu_tier_$1.log)" >> $L
}
way defaults "" "$SEL"
SMALL=${@:-tests --deselect tests/test_gpu_fullsize.py}
way lat2 "lat=2" "$SMALL"
way lat0 "lat=0" "$SMALL"
way novis0 "novis=0" "$SMALL"
cat $L
