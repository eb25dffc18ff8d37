#!/bin/bash
# A/B at C2 (1M x 128 f32, bitset graphs): HNSW_MI355X_NOVIS = 1 (visited bitsets, default there) / 2 (no visited set)
cd "$(dirname "$0")/.."
Q="--no-cpu-baseline --no-add-modes --no-clustered-check --no-process-warmup --recall-study-n 0 --steps 10"
for m in 1 2 1 2; do
  HNSW_MI355X_NOVIS=$m python bench.py $Q > gpurun_out/r4_novis_c2_$m.json 2> gpurun_out/r4_novis_c2_$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4_novis_c2_$m.json"))
print("c2 NOVIS=$m", d["value"], "q/s  resident", d["resident_queries_per_sec"], " ms/launch", d["roofline"]["avg_launch_us"]/1e3, " evals/query", d["evals_per_query"], " frac", d["roofline"]["frac"], " small", d["small_batch"]["queries_per_sec"], d["small_batch"]["resident_queries_per_sec"], " repeats", d["search_repeats"])
PY
done
