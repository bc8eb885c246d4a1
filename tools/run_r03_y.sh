#!/bin/bash
# (8,9) pair-row operations for small batches: parity (pair rows, pool, C++ API suite), small-batch numbers at the ipcl:: API
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03y
mkdir -p $OUT
cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_pair_rows.py tests/test_gpu_pool.py tests/test_gpu_cpp_api.py tests/test_gpu_paillier.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 ./pailliercryptolib_amd/ipcl_api_bench > $OUT/ipcl_api_bench.txt 2>&1; echo "api bench rc=$?"; head -14 $OUT/ipcl_api_bench.txt
