#!/bin/bash
# A/B of the search launches without visited sets (HNSW_MI355X_NOVIS) on the 10M int8 (C5-size) and 10M f32 (C4-size) indices
cd "$(dirname "$0")/.."
Q="--no-cpu-baseline --no-add-modes --small-batch 0 --no-clustered-check --no-process-warmup --recall-study-n 0 --steps 10"
for cfg in "c5 --metric sq_euclid_i8 --dim 96 --index-size 10000000 --nq 12500" "c4 --index-size 10000000 --nq 12500"; do
  set -- $cfg; name=$1; shift
  for m in 0 1; do
    HNSW_MI355X_NOVIS=$m python bench.py $Q "$@" > gpurun_out/r4_novis_${name}_$m.json 2> gpurun_out/r4_novis_${name}_$m.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r4_novis_${name}_$m.json"))
print("${name} NOVIS=$m", d["value"], "q/s  resident", d["resident_queries_per_sec"], " ms/launch", d["roofline"]["avg_launch_us"]/1e3, " evals/query", d["evals_per_query"], " recall", d["recall_at_10"], " repeats", d["search_repeats"])
PY
  done
done
