#!/bin/bash
# rocprofv3 evidence, round 5.  Kernel trace + stats and PMC passes, each in its own run (never mixed with trace domains):
#  f1   headline step, ONE batch in flight (a lone caller: launch durations are per kernel)
#  f2   headline step, two batches in flight (the round-4 headline: sequential-halves launches with CU claims side by side)
#  f4   headline step, default (FOUR batches in flight: one-lane product-scanning decrypts on a quarter of the chip each)
#  c4   config 4 (65536 x 3072-bit), c5 config 5 (1 M CT+CT / CT x PT)
# plus the 8-entry oversubscribed pool lines (host issue time), the API bench and the key-size table.
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --sustain-seconds 0"
RUNS=${RUNS:-"f1 f2 f4 c4 c5"}   # RUNS="c4" EXTRA=0: one configuration only, no second half
for run in $RUNS; do
  OUT=$REPO/gpurun_out/prof_r05$run
  mkdir -p $OUT
  case $run in
    f1) CMD="python $REPO/bench.py --in-flight 1 --steps 10 --warmup 2 $B";;
    f2) CMD="python $REPO/bench.py --in-flight 2 --steps 20 --warmup 3 $B";;
    f4) CMD="python $REPO/bench.py --in-flight 4 --steps 20 --warmup 3 $B";;
    c4) CMD="python $REPO/bench.py --config 4 --steps 3 --warmup 1 $B";;
    c5) CMD="python $REPO/bench.py --config 5 --steps 8 $B";;
  esac
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
  if [ $run = f1 ] || [ $run = f2 ] || [ $run = f4 ] || [ $run = c4 ]; then
    timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
    timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
  fi
  tail -2 $OUT/trace.log
done
cd $REPO
OUT=$REPO/gpurun_out/r05p
mkdir -p $OUT
if [ "${EXTRA:-1}" = 0 ]; then exit 0; fi
python bench.py --steps 20 --warmup 3 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py --in-flight 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_f1.json 2>/dev/null
python bench.py --in-flight 2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_f2.json 2>/dev/null
python bench.py --config 4 --steps 5 --warmup 1 > $OUT/bench_c4.json 2>/dev/null
python bench.py --config 5 --steps 12 > $OUT/bench_c5.json 2>/dev/null
BENCH_SINGLE_DEVICE=1 python bench.py --gpus 8 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --sustain-seconds 0 > $OUT/bench_n8_pool_1dev.json 2> $OUT/bench_n8.err
BENCH_SINGLE_DEVICE=1 python bench.py --gpus 8 --config 4 --steps 3 --warmup 1 > $OUT/bench_c4_n8_pool_1dev.json 2>/dev/null
BENCH_SINGLE_DEVICE=1 python bench.py --gpus 8 --config 5 --steps 8 > $OUT/bench_c5_n8_pool_1dev.json 2>/dev/null
./pailliercryptolib_amd/ipcl_api_bench > $OUT/ipcl_api_bench.txt 2>&1
for c in 16384 65536 131072; do python tools/bench_keysizes.py $c > $OUT/keysizes_$c.txt 2>&1; done
python tools/probe_lanes.py --count 65536 --lanes 1 --steps 6 > $OUT/big_ps1.txt 2>&1; python tools/probe_lanes.py --count 65536 --lanes 1 --steps 6 --ps 0 > $OUT/big_ps0.txt 2>&1
python tools/probe_lanes.py --lanes 1 2 3 4 --steps 40 > $OUT/lanes.txt 2>&1
python tools/probe_trace.py 2 8 > $OUT/trace_2lanes.txt 2>&1
python tools/probe_lanes.py --lanes 1 4 --steps 16 --gather 1 > $OUT/lanes_masked.txt 2>&1
python tools/probe_trace.py 4 20 > $OUT/trace_4lanes.txt 2>&1
(for w in 1 0; do for v in pageable pinned; do for c in 1 2 4; do echo "# PGPU_D2H_PRESYNC=$w  python tools/probe_two_callers.py $v $c 8"; PGPU_D2H_PRESYNC=$w python tools/probe_two_callers.py $v $c 8 2>&1 | grep -v amdgpu.ids | head -$((c+1)); done; done; done
 for v in pageable pinned; do echo "# PGPU_HOST_ADAPT=1  python tools/probe_two_callers.py $v 2 8"; PGPU_HOST_ADAPT=1 python tools/probe_two_callers.py $v 2 8 2>&1 | grep -v amdgpu.ids | head -3; done) > $OUT/two_callers.txt 2>&1
(for t in 1 2 3 4; do ./pailliercryptolib_amd/ipcl_api_bench --threads $t 8192 8 2>&1 | grep -v amdgpu.ids; done) > $OUT/ipcl_api_threads.txt 2>&1
(echo "# small batches, T host threads (each encrypt + decrypt through the ipcl:: API, vector in / out), 400 rounds: tests/cpp/ipcl_bench.cpp --threads T n 400"; for n in 64 700 2048; do for t in 1 2 4 8; do ./pailliercryptolib_amd/ipcl_api_bench --threads $t $n 400 2>&1 | grep -v amdgpu.ids; done; done) > $OUT/ipcl_api_threads_small.txt
ls $OUT
