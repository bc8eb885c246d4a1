#!/bin/bash
# r03 second run: pair-row parity, pool scenarios, API stage probe, bench (headline + config 5)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03b
mkdir -p $OUT
cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_pair_rows.py tests/test_gpu_pool.py -m gpu -x -q > $OUT/pytest_pair.log 2>&1; echo "pytest pair rc=$?"
tail -25 $OUT/pytest_pair.log
python3 -c "from pailliercryptolib_amd import build; build.build_api_bench()" > /dev/null 2>&1
g++ -O2 -std=c++17 -fopenmp -Iinclude -Itests/cpp -Ipailliercryptolib_amd/csrc/host tools/api_probe.cpp -Lpailliercryptolib_amd -lipcl_amd -lpgpu -Wl,-rpath,$REPO/pailliercryptolib_amd -o /tmp/api_probe 2> $OUT/probe_build.err
timeout 200 /tmp/api_probe > $OUT/api_probe.txt 2>&1; echo "probe rc=$?"
cat $OUT/api_probe.txt
timeout 400 python3 bench.py --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "n1 rc=$?"; tail -c 600 $OUT/bench_n1.err
timeout 400 python3 bench.py --config 5 --steps 8 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"; tail -c 600 $OUT/bench_c5.err
PGPU_PAIR_ROWS=0 timeout 400 python3 bench.py --config 5 --steps 8 > $OUT/bench_c5_words.json 2> $OUT/bench_c5_words.err; echo "c5 words rc=$?"
