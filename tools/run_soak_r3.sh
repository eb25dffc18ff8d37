#!/bin/bash
# Round-3 parity soak on the GPU box (final kernels: group windows, shadow traversals, wave-parallel heap pop): mid-size builds and
# large query sets against the oracle.  Output: gpurun_out/soak_r3.log (copied to profiles/r3_parity_soak.log).
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; L=$O/soak_r3.log; : > $L
run() { echo "== $*" >> $L; ( "$@" >> $L 2>&1 ) || echo "FAILED: $*" >> $L; }
run python3 tools/soak.py 100000 32 100000 uniform sq_euclid
run python3 tools/soak.py 100000 16 150000 uniform sq_euclid
run python3 tools/soak.py 60000 128 60000 clustered sq_euclid
run python3 tools/soak.py 60000 96 60000 uniform ucosine
run python3 tools/soak.py 30000 264 20000 uniform cosine
run python3 tools/soak.py 100000 96 100000 uniform sq_euclid_i8
HNSW_MI355X_VIS_HASH=1 run python3 tools/soak.py 60000 32 100000 uniform sq_euclid
HNSW_MI355X_SHADOW=0 run python3 tools/soak.py 60000 24 100000 uniform sq_euclid
run python3 tools/soak_window.py 40000
grep -c "DIFFERENT\|FAILED" $L; tail -40 $L
