#!/bin/bash
# Final lines of round 3 on the final build: the plain lines of the 10M configurations (exact-window cap 256), the same
# configurations at 65 536 queries per call, and the two multi-GPU rehearsals on the one GPU of the box.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles3; mkdir -p $O
C4="--index-size 10000000 --nq 12500"
C5="--metric sq_euclid_i8 --dim 96 --index-size 10000000 --nq 12500"
python3 $R/bench.py $C4 --small-batch 0 --steps 10 --recall-study-n 0 --no-clustered-check > $O/bench_c4_size.log 2>&1; echo c4 done
python3 $R/bench.py $C5 --small-batch 0 --steps 10 --recall-study-n 0 --no-clustered-check > $O/bench_c5_size.log 2>&1; echo c5 done
Q="--no-cpu-baseline --no-add-modes --small-batch 0 --no-clustered-check --no-process-warmup --recall-study-n 0"
python3 $R/bench.py $Q --index-size 10000000 --nq 65536 --steps 5 > $O/bench_c4_65536.log 2>&1; echo c4 65536 done
python3 $R/bench.py $Q --metric sq_euclid_i8 --dim 96 --index-size 10000000 --nq 65536 --steps 5 > $O/bench_c5_65536.log 2>&1; echo c5 65536 done
cd $R
BENCH_BACKEND=gloo BENCH_ALLOW_SHARED_GPUS=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --index-size 200000 --nq 8192 --steps 3 --warmup 1 $Q > $O/bench_2rank_gloo.log 2>&1; echo gloo done
BENCH_ALLOW_SHARED_GPUS=1 timeout -k 10 300 python3 bench.py --gpus 2 --sharding native --index-size 200000 --nq 8192 --steps 3 --warmup 1 --no-cpu-baseline --small-batch 0 --no-clustered-check --recall-study-n 0 > $O/bench_native2.log 2>&1; echo native done
for f in bench_c4_size bench_c5_size bench_c4_65536 bench_c5_65536 bench_2rank_gloo bench_native2; do echo $f; grep '^{' $O/$f.log | cut -c1-260; done
