#!/bin/bash
# r03: A/B-wavefront decrypt kernel: parity, then speed (one / two batches in flight; policy 0 / 1 / auto)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03i
mkdir -p $OUT
cd $REPO
PGPU_AB_DECRYPT=1 timeout 200 python3 bench.py --in-flight 1 --no-extras --no-cpu-baseline --steps 5 > $OUT/b_ab1_f1.json 2> $OUT/b_ab1_f1.err; echo "first AB run rc=$?"; tail -c 400 $OUT/b_ab1_f1.err
PGPU_AB_DECRYPT=1 timeout 600 python3 -m pytest tests/test_gpu_pair_rows.py tests/test_gpu_sha256_fullsize.py -m gpu -x -q -k "ab_wavefront or 2500" > $OUT/pytest_ab.log 2>&1; echo "pytest AB rc=$?"; tail -4 $OUT/pytest_ab.log
for pol in 0 1 2 3 0 2; do for fl in 1 2; do
  PGPU_AB_DECRYPT=$pol timeout 200 python3 bench.py --in-flight $fl --no-extras --no-cpu-baseline --steps 20 > $OUT/b_ab${pol}_f${fl}.json 2> $OUT/b_ab${pol}_f${fl}.err
  python3 -c "
import json
d=json.load(open('$OUT/b_ab${pol}_f${fl}.json')); print('policy $pol in-flight $fl', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('one_batch_in_flight',{}).get('ms_per_step'))"
done; done
