#!/bin/bash
cd "$(dirname "$0")/.."
Q="--no-cpu-baseline --no-add-modes --no-clustered-check --no-process-warmup --recall-study-n 0 --steps 5 --small-batch 0 --dim 768 --metric ucosine --max-edges 32 --ef-construction 400 --nq 32768"
for m in 0 2; do
  HNSW_MI355X_NOVIS=$m python bench.py $Q > gpurun_out/r4_c3_novis_$m.json 2> gpurun_out/r4_c3_novis_$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4_c3_novis_$m.json"))
print("c3 NOVIS=$m", d["value"], "q/s resident", d["resident_queries_per_sec"], " build", d["add_per_sec"], d["build_seconds"], " insert s", d["roofline_add"]["insert_search"]["seconds"], " ms/launch", d["roofline"]["avg_launch_us"]/1e3)
PY
done
