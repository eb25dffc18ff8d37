#!/bin/bash
# Round 5: A/B of the int8 search kernel's occupancy target on the C5-size index (10M x 96 int8, 12 500-query calls):
# product (5 waves per SIMD) against build variants -DHNSW_I8_WAVES=4 / 6 (build_variants/i8w4.so, i8w6.so), and the product with
# the shadow traversals off.  One bench line each into gpurun_out/r5_c5_ab.log.
O=$GRAFT_REPO_ROOT/gpurun_out; L=$O/r5_c5_ab.log; : > $L
A="--metric sq_euclid_i8 --dim 96 --index-size 10000000 --nq ${NQ:-12500} --insert-batch 65536 --no-cpu-baseline --no-add-modes --small-batch 0 --no-clustered-check --steps 10 --recall-queries 200"
run() { echo "== $1" >> $L; shift; ( "$@" python3 bench.py $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print(json.dumps({'queries_per_s': d['value'], 'resident': d['resident_queries_per_sec'], 'ms_per_step': d['ms_per_step'], 'launch_us': r['avg_launch_us'], 'frac': r['frac'], 'rows_per_launch': r['rows_measured_per_launch'], 'recall': d['recall_at_10'], 'repeats': d['search_repeats']}))" >> $L ) || echo FAILED >> $L; }
run "product (5 waves per SIMD)" env
run "shadows off" env HNSW_MI355X_DIAG=shadow=0
[ -f build_variants/i8w4.so ] && run "4 waves per SIMD" env HNSW_MI355X_LIB=$GRAFT_REPO_ROOT/build_variants/i8w4.so
[ -f build_variants/i8w6.so ] && run "6 waves per SIMD" env HNSW_MI355X_LIB=$GRAFT_REPO_ROOT/build_variants/i8w6.so
cat $L
