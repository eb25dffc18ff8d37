Q="--no-cpu-baseline --no-add-modes --small-batch 0 --no-clustered-check --no-process-warmup --recall-study-n 0"
L=hnswindex.net_amd/artifacts/native/linux-x64/HNSWIndex.Native.so
mkdir -p gpurun_out/exp
cp tools/_exp_variant.so $L
timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_index.py tests/test_gpu_exact_window.py -x -q -m gpu > gpurun_out/exp/tests.log 2>&1; tail -3 gpurun_out/exp/tests.log
HNSW_MI355X_SORTED_TOP=0 python3 tools/exp_tail2.py > gpurun_out/exp/alone_exact.log 2>&1; grep nq= gpurun_out/exp/alone_exact.log
for v in prod exp prod exp; do
cp tools/_${v}_variant.so $L
python3 bench.py $Q --steps 10 --nq 12500 > gpurun_out/exp/ab_c2s_$v.log 2>&1
python3 bench.py $Q --steps 10 --index-size 10000000 --nq 12500 > gpurun_out/exp/ab_c4_$v.log 2>&1
python3 bench.py $Q --steps 10 --metric sq_euclid_i8 --dim 96 --index-size 10000000 --nq 12500 > gpurun_out/exp/ab_c5_$v.log 2>&1
for f in gpurun_out/exp/ab_*_$v.log; do echo $f; grep '^{' $f | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d.get('resident_queries_per_sec'), r['avg_launch_us'], r['frac'], r.get('frac_of_measured_gather'), d.get('search_repeats'))"; done
done
cp tools/_exp_variant.so $L
