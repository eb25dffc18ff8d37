#!/bin/bash
# Runs on the GPU box (via gpurun): the plain bench line, kernel stats + PMC passes for bench.py with a
# FETCH_SIZE calibration on a known byte count in the same access pattern (tools/kbench), the host-traversal
# mode through the inner C ABI (slot_distance_kernel), and the other BASELINE configurations at full size.
# Summaries land in gpurun_out/profiles/; tools/install_profiles.py copies them into profiles/.
#   usage: tools/run_profiles.sh [core|configs|all]
set -o pipefail
WHAT=${1:-all}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O
[ -x $R/tools/kbench ] || hipcc --offload-arch=gfx950 -O3 -ffp-contract=off $R/tools/kbench.hip -o $R/tools/kbench
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-add-modes --small-batch 0 --no-clustered-check --no-process-warmup"
if [ $WHAT = core ] || [ $WHAT = all ]; then
# 0) the plain default line (no profiler attached)
python3 $R/bench.py > $O/bench_plain.log 2>&1
# 1) kernel trace + stats of the default bench command (query + build kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- python3 $R/bench.py $Q > $O/bench_under_rocprof.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_stats $O/kernel_stats.json hnsw:: > /dev/null
cp /tmp/p_stats/*kernel_stats.csv $O/ 2>/dev/null
# 2) PMC passes (separate runs, no tracing beyond what --pmc needs)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o f -- python3 $R/bench.py $Q --steps 2 --warmup 0 > $O/bench_pmc_fetch.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_fetch $O/pmc_fetch.json graph_search_kernel graph_insert_search_kernel graph_link_kernel > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o w -- python3 $R/bench.py $Q --steps 2 --warmup 0 > $O/bench_pmc_write.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_write $O/pmc_write.json graph_search_kernel graph_insert_search_kernel graph_link_kernel > /dev/null
# 3) calibration: kbench reads a known number of distinct random rows from a 2 GB matrix (> L3)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_cal -o c -- $R/tools/kbench 4000000 32768 21 0 > $O/kbench_calibration.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_cal $O/pmc_calibration.json v1 > /dev/null
# 4) the literal north-star split: traversal on the host, distances step by step through hnswdev_step_submit / _wait
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_host -o h -- python3 $R/bench.py --traversal host --n 200000 --nq 16384 --steps 3 --warmup 1 $Q > $O/bench_host_cabi.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_host $O/kernel_stats_host_cabi.json hnsw:: > /dev/null
fi
if [ $WHAT = configs ] || [ $WHAT = all ]; then
# 5) the other BASELINE configurations at full size, one GPU
python3 $R/bench.py --dim 768 --metric ucosine --max-edges 32 --ef-construction 400 --nq 32768 --small-batch 0 --steps 10 --recall-study-n 0 --seq-adds 500 --bounded-adds 1024 --batched-adds 8192 > $O/bench_c3.log 2>&1
python3 $R/bench.py --index-size 10000000 --nq 12500 --small-batch 0 --steps 10 --recall-study-n 0 > $O/bench_c4_size.log 2>&1
python3 $R/bench.py --metric sq_euclid_i8 --dim 96 --index-size 10000000 --nq 12500 --small-batch 0 --steps 10 --recall-study-n 0 > $O/bench_c5_size.log 2>&1
python3 $R/bench.py --data clustered > $O/bench_clustered.log 2>&1
fi
ls -la $O
for f in $O/bench_*.log; do echo $f; grep '^{' $f | cut -c1-400; done
