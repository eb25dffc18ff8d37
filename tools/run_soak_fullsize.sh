#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; L=$O/soak_fullsize_r3.log; : > $L
python3 tools/soak_fullsize.py 1000000 128 sq_euclid 262144 65536 >> $L 2>&1 || echo FAILED >> $L
python3 tools/soak_fullsize.py 1000000 128 sq_euclid 50000 12500 >> $L 2>&1 || echo FAILED >> $L
python3 tools/soak_fullsize.py 10000000 128 sq_euclid 50000 12500 >> $L 2>&1 || echo FAILED >> $L
python3 tools/soak_fullsize.py 10000000 96 sq_euclid_i8 25000 12500 >> $L 2>&1 || echo FAILED >> $L
grep -v amdgpu $L
