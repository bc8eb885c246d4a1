#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03f
mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_pair_rows.py tests/test_gpu_cpp_api.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 300 ./pailliercryptolib_amd/ipcl_api_bench > $OUT/ipcl_api_bench.txt 2>&1; echo "api bench rc=$?"; head -8 $OUT/ipcl_api_bench.txt
g++ -O2 -std=c++17 -fopenmp -Iinclude -Itests/cpp -Ipailliercryptolib_amd/csrc/host tools/api_probe.cpp -Lpailliercryptolib_amd -lipcl_amd -lpgpu -Wl,-rpath,$REPO/pailliercryptolib_amd -o /tmp/api_probe 2> $OUT/probe_build.err
timeout 200 /tmp/api_probe > $OUT/api_probe.txt 2>&1; cat $OUT/api_probe.txt
IPCL_MALLOC_TUNING=0 timeout 200 /tmp/api_probe 2>&1 | head -4
timeout 400 python3 bench.py --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err; python3 -c "
import json; d=json.load(open('$OUT/bench_n1.json')); print(d['value'], d['ms_per_step'], d['api_level'], d['masked_table_gather'])"
