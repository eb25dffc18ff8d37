#!/bin/bash
# Round-4 parity soak on the GPU box, final kernels of the round (search launches without a visited set, MFMA entry block,
# latency variants with the straight-line pool): mid-size builds and large query sets against the oracle, once more with the
# latency variants forced and once with the visited sets kept, then the full-size legs.  Output: gpurun_out/soak_r4.log and
# gpurun_out/soak_fullsize_r4.log (copied to profiles/r4_parity_soak.log / r4_soak_fullsize.log).
#   usage: tools/run_soak_r4.sh [mid|full]
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
if [ "${1:-mid}" = mid ]; then
L=$O/soak_r4.log; : > $L
run() { echo "== $*" >> $L; ( "$@" >> $L 2>&1 ) || echo "FAILED: $*" >> $L; }
run python3 tools/soak.py 100000 32 100000 uniform sq_euclid
run python3 tools/soak.py 100000 16 150000 uniform sq_euclid
run python3 tools/soak.py 60000 128 60000 clustered sq_euclid
run python3 tools/soak.py 60000 96 60000 uniform ucosine
run python3 tools/soak.py 30000 264 20000 uniform cosine
run python3 tools/soak.py 100000 96 100000 uniform sq_euclid_i8
echo "== HNSW_MI355X_LAT=2" >> $L
HNSW_MI355X_LAT=2 run python3 tools/soak.py 60000 32 60000 uniform sq_euclid
HNSW_MI355X_LAT=2 run python3 tools/soak.py 40000 96 40000 uniform ucosine
echo "== HNSW_MI355X_NOVIS=0" >> $L
HNSW_MI355X_NOVIS=0 run python3 tools/soak.py 60000 24 100000 uniform sq_euclid
echo "== HNSW_MI355X_VIS_HASH=1" >> $L
HNSW_MI355X_VIS_HASH=1 run python3 tools/soak.py 60000 32 100000 uniform sq_euclid
run python3 tools/soak_window.py 40000
echo "== HNSW_MI355X_LAT=0 (window)" >> $L
HNSW_MI355X_LAT=0 run python3 tools/soak_window.py 20000
grep -c "DIFFERENT\|FAILED" $L; grep -v amdgpu $L | tail -40
else
L=$O/soak_fullsize_r4.log; : > $L
python3 tools/soak_fullsize.py 1000000 128 sq_euclid 262144 65536 >> $L 2>&1 || echo FAILED >> $L
python3 tools/soak_fullsize.py 1000000 128 sq_euclid 50000 12500 >> $L 2>&1 || echo FAILED >> $L
python3 tools/soak_fullsize.py 10000000 128 sq_euclid 50000 12500 >> $L 2>&1 || echo FAILED >> $L
python3 tools/soak_fullsize.py 10000000 96 sq_euclid_i8 25000 12500 >> $L 2>&1 || echo FAILED >> $L
grep -v amdgpu $L
fi
