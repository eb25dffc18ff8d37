#!/bin/bash
# A/B of the 1M build with and without visited sets in Add's searches (HNSW_MI355X_NOVIS_INSERT)
cd "$(dirname "$0")/.."
Q="--no-cpu-baseline --no-add-modes --no-clustered-check --no-process-warmup --recall-study-n 0 --steps 3 --small-batch 0"
for m in 0 1 0 1; do
  HNSW_MI355X_NOVIS_INSERT=$m python bench.py $Q > gpurun_out/r4_novis_build_$m.json 2> gpurun_out/r4_novis_build_$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4_novis_build_$m.json"))
ra=d["roofline_add"]
print("build NOVIS_INSERT=$m", d["add_per_sec"], "adds/s", d["build_seconds"], "s  insert", ra["insert_search"]["seconds"], "s link", ra["link_half"]["seconds"], " evals", ra["evals"], " q/s", d["value"])
PY
done
