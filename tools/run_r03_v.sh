#!/bin/bash
# 4096-bit key class in split form / pair rows: parity, then per-key-class timings with the split form on / off
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03v
mkdir -p $OUT
cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_pair_rows.py tests/test_gpu_hensel.py tests/test_gpu_paillier.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 900 python3 tools/bench_keysizes.py 16384 > $OUT/keysizes.txt 2>&1; echo "keysizes rc=$?"; cat $OUT/keysizes.txt | tail -10
