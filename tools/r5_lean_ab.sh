#!/bin/bash
# Round 5: A/B of the LEAN search kernel (no visited set and the overlapped form as compile-time facts, the exact two-heap traversal as a
# real call instead of inlined code) at several occupancy targets, against the product library.  Variants: build_variants/lc_i<W>_f<F>.so
# (-DHNSW_EXP_LEAN=true -DHNSW_COLD_EXACT -DHNSW_I8_WAVES=W -DHNSW_F32_WAVES=F).  One summary line per run into gpurun_out/r5_lean_ab.log.
#   usage: tools/r5_lean_ab.sh [c5] [c5L] [c4] [c2]
O=$GRAFT_REPO_ROOT/gpurun_out; L=$O/r5_lean_ab.log; mkdir -p $O; : > $L
Q="--no-cpu-baseline --no-add-modes --small-batch 0 --no-clustered-check --steps 10 --recall-queries 200 --insert-batch 65536 --recall-study-n 0"
C5="--metric sq_euclid_i8 --dim 96 --index-size 10000000 --nq 12500"
C5L="--metric sq_euclid_i8 --dim 96 --index-size 10000000 --nq 65536"
C4="--index-size 10000000 --nq 12500"
C2=""
run() { # label, lib ('' = product), bench args...
  local label=$1 lib=$2; shift 2
  echo "== $label" >> $L
  ( case "$lib" in diag:*) export HNSW_MI355X_DIAG="${lib#diag:}";; ?*) export HNSW_MI355X_LIB=$GRAFT_REPO_ROOT/build_variants/$lib;; esac
    timeout -k 10 300 python3 bench.py $Q "$@" 2>$O/r5_lean_ab.err | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print(json.dumps({'queries_per_s': d['value'], 'resident': d['resident_queries_per_sec'], 'ms_per_step': d['ms_per_step'], 'launch_us': r['avg_launch_us'], 'frac': r['frac'], 'rows_per_launch': r['rows_measured_per_launch'], 'recall': d['recall_at_10'], 'repeats': d['search_repeats'], 'build_id': d['build_id'][:16], 'add_per_sec': d.get('add_per_sec'), 'insert_search_frac': (d.get('roofline_add') or {}).get('insert_search', {}).get('frac'), 'link_frac': (d.get('roofline_add') or {}).get('link_half', {}).get('frac')}))" >> $L ) || { echo FAILED >> $L; tail -3 $O/r5_lean_ab.err >> $L; }
  echo "$label done"
}
for what in ${@:-c5 c5L c4 c2}; do
case $what in
c5)  for v in ${VARIANTS:-"" lc_i5_f3.so lc_i6_f4.so lc_i7_f3.so lc_i8_f3.so}; do [ "$v" = product ] && v="";  [ -z "$v" ] || [ "${v#diag:}" != "$v" ] || [ -f build_variants/$v ] && run "C5-size 12500 ${v:-product}" "$v" $C5; done;;
c5L) for v in ${VARIANTS:-"" lc_i6_f4.so lc_i8_f3.so}; do [ "$v" = product ] && v="";  [ -z "$v" ] || [ "${v#diag:}" != "$v" ] || [ -f build_variants/$v ] && run "C5-size 65536 ${v:-product}" "$v" $C5L; done;;
c4)  for v in ${VARIANTS:-"" lc_i5_f3.so lc_i6_f4.so}; do [ "$v" = product ] && v="";  [ -z "$v" ] || [ "${v#diag:}" != "$v" ] || [ -f build_variants/$v ] && run "C4-size 12500 ${v:-product}" "$v" $C4; done;;
c2)  for v in ${VARIANTS:-"" lc_i5_f3.so lc_i6_f4.so}; do [ "$v" = product ] && v="";  [ -z "$v" ] || [ "${v#diag:}" != "$v" ] || [ -f build_variants/$v ] && run "C2 65536 ${v:-product}" "$v" $C2; done;;
esac
done
cat $L
