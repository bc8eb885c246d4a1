#!/bin/bash
# rocprofv3 evidence, round 6.  Kernel trace + stats and PMC passes, each in its own run (never mixed with trace domains):
#  f4   headline step, default (FOUR batches in flight: one-lane product-scanning decrypts on a quarter of the chip each)
#  f1   headline step, ONE batch in flight (a lone caller: launch durations are per kernel)
#  c4   config 4 (65536 x 3072-bit), c5 config 5 (1 M CT+CT / CT x PT)
# then the bench lines themselves (compact line -> *.line.json, detail record -> *.json) and the API bench.
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --sustain-seconds 0"
RUNS=${RUNS:-"f4 f1 c4 c5"}
for run in $RUNS; do
  OUT=$REPO/gpurun_out/prof_r06$run
  mkdir -p $OUT
  case $run in
    f1) CMD="python $REPO/bench.py --in-flight 1 --steps 10 --warmup 2 $B";;
    f4) CMD="python $REPO/bench.py --in-flight 4 --steps 20 --warmup 3 $B";;
    c4) CMD="python $REPO/bench.py --config 4 --steps 3 --warmup 1 $B";;
    c5) CMD="python $REPO/bench.py --config 5 --steps 8 $B";;
  esac
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
  timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
  timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
  timeout 500 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
  tail -2 $OUT/trace.log
done
cd $REPO
OUT=$REPO/gpurun_out/r06p
mkdir -p $OUT
if [ "${EXTRA:-1}" = 0 ]; then exit 0; fi
line() { python bench.py "${@:2}" > $OUT/$1.line.json 2> $OUT/$1.err; cp gpurun_out/bench_detail.json $OUT/$1.json; }
line bench_n1 --steps 20 --warmup 5
line bench_f1 --in-flight 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras
line bench_f2 --in-flight 2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras
line bench_c4 --config 4 --steps 5 --warmup 1
line bench_c5 --config 5 --steps 12
BENCH_SINGLE_DEVICE=1 line bench_n8_pool_1dev --gpus 8 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --sustain-seconds 0
./pailliercryptolib_amd/ipcl_api_bench > $OUT/ipcl_api_bench.txt 2>&1
python tools/bench_small_cpu.py > $OUT/small_batch_cpu_ifma.txt 2>&1
(for b in 2048 3072 1024; do python tools/bench_wave_sizes.py $b; done; echo '# PGPU_WAVE_WIDEQ=0 (masked quotient digits)'; PGPU_WAVE_WIDEQ=0 python tools/bench_wave_sizes.py 2048 16 512) 2>&1 | grep -v amdgpu.ids > $OUT/wave_form_sizes.txt
for c in 16384 65536; do python tools/bench_keysizes.py $c > $OUT/keysizes_$c.txt 2>&1; done
(for t in 1 2 4; do ./pailliercryptolib_amd/ipcl_api_bench --threads $t 8192 8 2>&1 | grep -v amdgpu.ids; done) > $OUT/ipcl_api_threads.txt 2>&1
ls $OUT
