#!/bin/bash
# sequential-halves DJN encrypt: parity, config 4 with the form on / off, 2048-bit encrypt of 32768
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03u
mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_pair_rows.py -m gpu -x -q -k "sequential" > $OUT/pytest_seq.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_seq.log
for pol in 0 1 0 1; do
  PGPU_SEQ_DECRYPT=$pol timeout 600 python3 bench.py --config 4 --steps 3 --warmup 1 > $OUT/c4_seq$pol.json 2> $OUT/c4_seq$pol.err
  python3 -c "
import json
d=json.load(open('$OUT/c4_seq$pol.json')); r=d['roofline']; print('config4 seq=$pol', d['value'], d['ms_per_step'], r['kernel_ms'], r['other_kernels'])"
done
