#!/bin/bash
# A/B of the MFMA entry block (HNSW_MI355X_MFMA_ENTRY) at C2: resident query sets and 12 500-query calls
cd "$(dirname "$0")/.."
Q="--no-cpu-baseline --no-add-modes --no-clustered-check --no-process-warmup --recall-study-n 0 --steps 10"
for m in 0 1 0 1; do
  HNSW_MI355X_MFMA_ENTRY=$m python bench.py $Q > gpurun_out/r4_entry_$m.json 2> gpurun_out/r4_entry_$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4_entry_$m.json"))
print("MFMA_ENTRY=$m export", d["value"], " resident", d["resident_queries_per_sec"], " 12500/call", d["small_batch"]["queries_per_sec"], d["small_batch"]["resident_queries_per_sec"], " evals/query", d["evals_per_query"])
PY
done
