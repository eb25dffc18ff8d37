#!/bin/bash
# Runs on the GPU box (via gpurun): kernel stats + PMC passes for bench.py, with a FETCH_SIZE
# calibration on a known byte count in the same access pattern (tools/kbench).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O
[ -x $R/tools/kbench ] || hipcc --offload-arch=gfx950 -O3 -ffp-contract=off $R/tools/kbench.hip -o $R/tools/kbench
cd /tmp && export TMPDIR=/tmp
# 1) kernel trace + stats of the default bench command
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- python3 $R/bench.py --no-cpu-baseline --small-batch 0 > $O/bench_under_rocprof.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_stats $O/kernel_stats.json hnsw:: > /dev/null
cp /tmp/p_stats/*kernel_stats.csv $O/ 2>/dev/null
# 2) PMC passes (separate runs, no tracing beyond what --pmc needs)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --small-batch 0 --steps 2 --warmup 0 > $O/bench_pmc_fetch.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_fetch $O/pmc_fetch.json graph_search_kernel > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o w -- python3 $R/bench.py --no-cpu-baseline --small-batch 0 --steps 2 --warmup 0 > $O/bench_pmc_write.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_write $O/pmc_write.json graph_search_kernel > /dev/null
# 3) calibration: kbench reads a known number of distinct random rows from a 2 GB matrix (> L3)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_cal -o c -- $R/tools/kbench 4000000 32768 21 0 > $O/kbench_calibration.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p_cal $O/pmc_calibration.json v1 > /dev/null
ls -la $O
tail -1 $O/bench_under_rocprof.log | cut -c1-600
