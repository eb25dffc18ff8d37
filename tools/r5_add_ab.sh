#!/bin/bash
# Round 5: A/B of the Add paths (default build under the host-thread cap, B = 1, exact window, the bounded ladder, one 32 768-item snapshot) and of single-query calls
# between the product library and variants.  VARIANTS="product <file>.so diag:<string> ..." as in tools/r5_lean_ab.sh; summary lines -> gpurun_out/r5_add_ab.log
O=$GRAFT_REPO_ROOT/gpurun_out; L=$O/r5_add_ab.log; mkdir -p $O; : > $L
A="--no-clustered-check --recall-study-n 0 --small-batch 0 --steps 5 --recall-queries 200 $EXTRA"
for v in ${VARIANTS:-product}; do
  echo "== $v" >> $L
  ( case "$v" in product) ;; diag:*) export HNSW_MI355X_DIAG="${v#diag:}";; *) export HNSW_MI355X_LIB=$GRAFT_REPO_ROOT/build_variants/$v;; esac
    timeout -k 10 400 python3 bench.py $A 2>$O/r5_add_ab.err | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); m=d.get('add_modes',{}); ra=d.get('roofline_add',{})
o={'build_id': d['build_id'][:16], 'add_per_sec': d.get('add_per_sec'), 'build_insert_kernel_s': ra.get('insert_search',{}).get('seconds'), 'build_link_s': ra.get('link_half',{}).get('seconds'),
   'sequential': m.get('sequential',{}).get('adds_per_sec'), 'exact_window': m.get('exact_window',{}).get('adds_per_sec'),
   'ladder': {k: v.get('adds_per_sec') for k, v in m.get('bounded',{}).items() if k.startswith('B') and isinstance(v, dict)},
   'batched': m.get('batched',{}).get('adds_per_sec'), 'batched_roofline': (m.get('batched',{}).get('roofline') or {}).get('frac'),
   'one_query_ms': d.get('crossover_batch_vs_cpu',{}).get('one_query_call',{}).get('ms_per_call'), 'one_query_kernel_ms': d.get('crossover_batch_vs_cpu',{}).get('one_query_call',{}).get('kernel_ms'),
   'queries_per_s': d['value'], 'graph': d.get('build_evals')}
print(json.dumps(o))" >> $L ) || { echo FAILED >> $L; grep -v amdgpu.ids $O/r5_add_ab.err | tail -8 >> $L; }
  echo "$v done"
done
cat $L
