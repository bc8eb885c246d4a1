#!/bin/bash
# r03 first run: the whole GPU suite, the bench line, the instruction-phase microbenchmark, the host-glue timings.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03a
mkdir -p $OUT
cd $REPO
timeout 120 ./tools/ubench_phase > $OUT/ubench_phase.txt 2>&1; echo "phase rc=$?"
g++ -O2 -std=c++17 -fopenmp -Iinclude -Ipailliercryptolib_amd/csrc/host tools/host_glue_bench.cpp -Lpailliercryptolib_amd -lipcl_amd -lpgpu -Wl,-rpath,$REPO/pailliercryptolib_amd -o /tmp/hg 2> $OUT/hg_build.err
for t in 1 2 4 8 16; do echo "IPCL_NUM_THREADS=$t"; IPCL_NUM_THREADS=$t timeout 120 /tmp/hg; done > $OUT/host_glue.txt 2>&1; echo "hg rc=$?"
timeout 400 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "n1 rc=$?"
tail -c 1500 $OUT/bench_n1.err
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest_gpu.log
for w in 3 4; do PGPU_FIXED_WINDOW=$w timeout 200 python3 bench.py --no-extras --no-cpu-baseline --steps 10 > $OUT/bench_window$w.json 2> $OUT/bench_window$w.err; echo "w$w rc=$?"; done
