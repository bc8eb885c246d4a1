#!/bin/bash
# rocprofv3 evidence, round 3: kernel trace + stats and PMC passes (each in its own run; never mixed with trace domains)
# for (a) the headline step with ONE batch in flight (launch durations are then per kernel), (b) the same with two in
# flight (the default of bench.py), (c) config 5 (CT+CT on 1 M pair rows).  usage: profile_r03.sh
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for run in r03 r03c5; do
  OUT=$REPO/gpurun_out/prof_$run
  mkdir -p $OUT
  if [ $run = r03 ]; then CMD="python $REPO/bench.py --in-flight 1 --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
  else CMD="python $REPO/bench.py --config 5 --steps 6 --no-cpu-baseline --no-extras"; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
  tail -2 $OUT/trace.log
done
OUT=$REPO/gpurun_out/prof_r03f2
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py --in-flight 2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
