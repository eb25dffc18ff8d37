#!/bin/bash
# A/B of the exact-window / B=1 Add on the 1M index: HNSW_MI355X_TWO_LANE = 0 / 1 (round 4)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for mode in 0 1; do
  echo "== TWO_LANE=$mode" | tee -a gpurun_out/r4_ab_window.log
  HNSW_MI355X_TWO_LANE=$mode python tools/exact_window_bench.py 1000000 6000 64,256 128 2>&1 | tee -a gpurun_out/r4_ab_window.log
done
