#!/bin/bash
# Parity soak on the GPU box, final kernels of the round: mid-size builds and large query sets against the oracle under the defaults and
# under forced traversal forms, the exact window, tie-heavy fuzz cases with fresh seeds, then the full-size legs.
# Output: gpurun_out/soak_mid.log / soak_fuzz.log / soak_fullsize.log (copied to profiles/r<N>_parity_soak.log, _fuzz_more.log, _soak_fullsize.log).
#   usage: tools/run_soak.sh [mid|fuzz|full]
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
case "${1:-mid}" in
mid)
L=$O/soak_mid.log; : > $L
run() { echo "== $* (HNSW_MI355X_DIAG='$HNSW_MI355X_DIAG')" >> $L; ( "$@" >> $L 2>&1 ) || echo "FAILED: $*" >> $L; }
run python3 tools/soak.py 100000 32 100000 uniform sq_euclid
run python3 tools/soak.py 100000 16 150000 uniform sq_euclid
run python3 tools/soak.py 60000 128 60000 clustered sq_euclid
run python3 tools/soak.py 60000 96 60000 uniform ucosine
run python3 tools/soak.py 30000 264 20000 uniform cosine
run python3 tools/soak.py 100000 96 100000 uniform sq_euclid_i8
SOAK_CAP=256 run python3 tools/soak.py 60000 48 60000 uniform sq_euclid
SOAK_CAP=64 run python3 tools/soak.py 30000 32 30000 uniform ucosine
export HNSW_MI355X_DIAG="lat=2"
run python3 tools/soak.py 60000 32 60000 uniform sq_euclid
run python3 tools/soak.py 40000 96 40000 uniform ucosine
export HNSW_MI355X_DIAG="novis=0"
run python3 tools/soak.py 60000 24 100000 uniform sq_euclid
export HNSW_MI355X_DIAG="vis_hash=1"
run python3 tools/soak.py 60000 32 100000 uniform sq_euclid
export HNSW_MI355X_DIAG="sorted_top=0"
run python3 tools/soak.py 30000 32 30000 uniform sq_euclid
export HNSW_MI355X_DIAG=""
run python3 tools/soak_window.py 40000
export HNSW_MI355X_DIAG="lat=0"
run python3 tools/soak_window.py 20000
grep -c "DIFFERENT\|FAILED" $L; grep -v amdgpu $L | tail -40 ;;
fuzz)
L=$O/soak_fuzz.log; : > $L
python3 tools/fuzz_more.py 12000 360 >> $L 2>&1 || echo FAILED >> $L
HNSW_MI355X_DIAG="lat=2" python3 tools/fuzz_more.py 13000 120 >> $L 2>&1 || echo FAILED >> $L
grep -v amdgpu $L | tail -12 ;;
full)
L=$O/soak_fullsize.log; : > $L
python3 tools/soak_fullsize.py 1000000 128 sq_euclid 262144 65536 >> $L 2>&1 || echo FAILED >> $L
python3 tools/soak_fullsize.py 1000000 128 sq_euclid 50000 12500 >> $L 2>&1 || echo FAILED >> $L
python3 tools/soak_fullsize.py 10000000 128 sq_euclid 50000 12500 >> $L 2>&1 || echo FAILED >> $L
python3 tools/soak_fullsize.py 10000000 96 sq_euclid_i8 25000 12500 >> $L 2>&1 || echo FAILED >> $L
grep -v amdgpu $L ;;
esac
