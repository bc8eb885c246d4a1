#!/bin/bash
# r02 GPU run A: baseline sanity + lone-wave cadence microbenchmark (+ PMC view of it)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02a
mkdir -p $OUT
cd $REPO
./tools/ubench_lone > $OUT/ubench_lone.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_base.json 2> $OUT/bench_base.err
cd /tmp && export TMPDIR=/tmp
for v in 0 8; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_v$v -- $REPO/tools/ubench_lone $v > $OUT/pmc_v$v.log 2>&1
done
python $REPO/tools/pmc_kernel.py $OUT/pmc_v0 "k<0, 0>" > $OUT/pmc_summary.txt 2>&1
python $REPO/tools/pmc_kernel.py $OUT/pmc_v8 "k<8, 0>" >> $OUT/pmc_summary.txt 2>&1
cat $OUT/ubench_lone.txt $OUT/pmc_summary.txt; tail -c 600 $OUT/bench_base.json
