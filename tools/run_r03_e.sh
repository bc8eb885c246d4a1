#!/bin/bash
# r03: selective alignment pass + gather policy: new tests, bench with extras, small-batch API bench, profiles
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03e
mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_pair_rows.py -m gpu -x -q -k "gather or lanes or switched" > $OUT/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -5 $OUT/pytest_new.log
timeout 600 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "n1 rc=$?"; tail -c 300 $OUT/bench_n1.err
timeout 300 python3 bench.py --in-flight 1 --no-extras --no-cpu-baseline > $OUT/bench_f1.json 2> $OUT/bench_f1.err
timeout 300 ./pailliercryptolib_amd/ipcl_api_bench > $OUT/ipcl_api_bench.txt 2>&1; echo "api bench rc=$?"
bash tools/profile_r03.sh > $OUT/profile.log 2>&1; echo "profile rc=$?"
