#!/bin/bash
# Round 5 profile set, run on the GPU box (via gpurun).  Summaries land in gpurun_out/profiles5/; tools/install_profiles_r5.py
# copies them into profiles/ and derives r5_pmc_traffic.json (per configuration, stamped with the library's build id: bench.py quotes
# a traffic figure only for the build it was measured on), r5_gather_ceilings.json (random gathers from a table >> cache AND from a
# table of each configuration's own size), r5_issue_counters.json and r5_mfma_utilisation.json.
# The 10M / 768-d configurations build with the opt-in 65 536-item snapshots (--insert-batch 65536): under the default cap (the host's
# hardware threads) a 10M build takes minutes, and these passes measure the QUERY kernel.
#   usage: tools/run_profiles_r5.sh [c2|calib|c3|c4|c5|issue|misc|plain|all]
set -o pipefail
WHAT=${1:-all}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles5; mkdir -p $O
[ -x $R/tools/gather_bench ] || hipcc --offload-arch=gfx950 -O3 $R/tools/gather_bench.hip -o $R/tools/gather_bench
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-add-modes --small-batch 0 --no-clustered-check --no-process-warmup"
C3="--dim 768 --metric ucosine --max-edges 32 --ef-construction 400 --nq 32768 --insert-batch 65536"
C4="--index-size 10000000 --nq 12500 --insert-batch 65536"
C5="--metric sq_euclid_i8 --dim 96 --index-size 10000000 --nq 12500 --insert-batch 65536"
C5L="--metric sq_euclid_i8 --dim 96 --index-size 10000000 --nq 65536 --insert-batch 65536"
pmc() { # name, counters, bench args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --output-format csv -d /tmp/p_$name -o p -- python3 $R/bench.py $Q --steps 2 --warmup 0 "$@" > $O/bench_$name.log 2>&1
  python3 $R/tools/prof_summary.py /tmp/p_$name $O/$name.json graph_search_kernel graph_insert_search_kernel graph_link_kernel > /dev/null
  echo "$name done"; rm -rf /tmp/p_$name
}
if [ $WHAT = plain ]; then # only the plain bench lines of every configuration (no profiler attached)
python3 $R/bench.py > $O/bench_plain.log 2>&1; echo c2 plain done
python3 $R/bench.py $C4 --small-batch 0 --steps 10 --recall-study-n 0 --no-clustered-check > $O/bench_c4_size.log 2>&1; echo c4 plain done
python3 $R/bench.py $C5 --small-batch 0 --steps 10 --recall-study-n 0 --no-clustered-check > $O/bench_c5_size.log 2>&1; echo c5 plain done
python3 $R/bench.py $C3 --small-batch 0 --steps 10 --recall-study-n 0 --seq-adds 500 --window-adds 2000 --bounded-adds 1024 --batched-adds 8192 --no-clustered-check > $O/bench_c3.log 2>&1; echo c3 plain done
python3 $R/bench.py --data clustered > $O/bench_clustered.log 2>&1; echo clustered done
fi
if [ $WHAT = c2 ] || [ $WHAT = all ]; then
python3 $R/bench.py > $O/bench_plain.log 2>&1; echo plain done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- python3 $R/bench.py $Q > $O/bench_under_rocprof.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_stats $O/kernel_stats.json hnsw:: > /dev/null
cp /tmp/p_stats/*kernel_stats.csv $O/ 2>/dev/null; cp /tmp/p_stats/*/*kernel_stats.csv $O/ 2>/dev/null; rm -rf /tmp/p_stats
pmc c2_fetch FETCH_SIZE
pmc c2_write WRITE_SIZE
fi
if [ $WHAT = calib ] || [ $WHAT = all ]; then
for rb in 128 512 3072; do
  $R/tools/gather_bench $rb 4 16777216 0 > $O/gather_$rb.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_cal$rb -o c -- $R/tools/gather_bench $rb 4 16777216 0 > $O/gather_pmc_$rb.log 2>&1
  python3 $R/tools/prof_summary.py /tmp/p_cal$rb $O/gather_pmc_$rb.json gather_ > /dev/null; rm -rf /tmp/p_cal$rb
done
$R/tools/gather_bench 128 4 16777216 1 > $O/gather_128_v1.log 2>&1
# ... and from a table of each configuration's OWN size (uniform random rows; a table that half-fits the 256-MiB Infinity Cache is served partly from it)
$R/tools/gather_bench 512 0.476837 16777216 0 > $O/gather_own_c2.log 2>&1      # C2: 1M x 512 B
$R/tools/gather_bench 3072 2.861023 4194304 0 > $O/gather_own_c3.log 2>&1     # C3: 1M x 3072 B
$R/tools/gather_bench 512 4.768372 16777216 0 > $O/gather_own_c4.log 2>&1     # C4: 10M x 512 B
$R/tools/gather_bench 128 1.192093 16777216 0 > $O/gather_own_c5.log 2>&1     # C5: 10M x 128 B
echo calib done
fi
if [ $WHAT = c3 ] || [ $WHAT = all ]; then
python3 $R/bench.py $C3 --small-batch 0 --steps 10 --recall-study-n 0 --seq-adds 500 --window-adds 2000 --bounded-adds 1024 --batched-adds 8192 --no-clustered-check > $O/bench_c3.log 2>&1; echo c3 plain done
pmc c3_fetch FETCH_SIZE $C3
pmc c3_write WRITE_SIZE $C3
pmc c3_mfma "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $C3
fi
if [ $WHAT = c4 ] || [ $WHAT = all ]; then
python3 $R/bench.py $C4 --small-batch 0 --steps 10 --recall-study-n 0 --no-clustered-check > $O/bench_c4_size.log 2>&1; echo c4 plain done
pmc c4_fetch FETCH_SIZE $C4
pmc c4_write WRITE_SIZE $C4
fi
if [ $WHAT = c5 ] || [ $WHAT = all ]; then
python3 $R/bench.py $C5 --small-batch 0 --steps 10 --recall-study-n 0 --no-clustered-check > $O/bench_c5_size.log 2>&1; echo c5 plain done
pmc c5_fetch FETCH_SIZE $C5
pmc c5_write WRITE_SIZE $C5
fi
if [ $WHAT = misc ] || [ $WHAT = all ]; then
python3 $R/bench.py --data clustered > $O/bench_clustered.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_host -o h -- python3 $R/bench.py --traversal host --n 200000 --nq 16384 --steps 3 --warmup 1 $Q > $O/bench_host_cabi.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_host $O/kernel_stats_host_cabi.json hnsw:: > /dev/null; rm -rf /tmp/p_host
echo misc done
fi
if [ $WHAT = issue ] || [ $WHAT = all ]; then # what bounds the traversals now: instruction issue against wave cycles on the PRODUCT kernel (one pass per counter group)
GA="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
GB="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
for cfg in ${ISSUE_CFGS:-c5 c5L c4 c2}; do
  case $cfg in c5) A="$C5";; c5L) A="$C5L";; c4) A="$C4";; c2) A="";; esac
  pmc ${cfg}_issue_a "$GA" $A
  pmc ${cfg}_issue_b "$GB" $A
  if [ $cfg = c5L ]; then pmc c5L_fetch FETCH_SIZE $A; pmc c5L_write WRITE_SIZE $A; fi
done
fi
if [ $WHAT = pmc_rest ]; then # only the PMC traffic passes of the 10M / 768-d configurations (after a change of the library: bench.py quotes traffic per build id)
pmc c4_fetch FETCH_SIZE $C4
pmc c4_write WRITE_SIZE $C4
pmc c5_fetch FETCH_SIZE $C5
pmc c5_write WRITE_SIZE $C5
pmc c3_fetch FETCH_SIZE $C3
pmc c3_write WRITE_SIZE $C3
fi
if [ $WHAT = icache ]; then # instruction-cache behaviour of the two big kernels (search launches and the 1M build)
pmc c2_icache "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
pmc c5_icache "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" $C5
fi
ls -la $O
for f in $O/bench_*.log; do echo $f; grep '^{' $f | cut -c1-300; done
