#!/bin/bash
# sequential-halves CT x PT kernel: parity, then config 5 with it on / off
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03r
mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_pair_rows.py -m gpu -x -q -k "sequential" > $OUT/pytest_seq.log 2>&1; echo "pytest seq rc=$?"; tail -4 $OUT/pytest_seq.log
for pol in 0 1; do
  PGPU_SEQ_DECRYPT=$pol timeout 600 python3 bench.py --config 5 --steps 3 --warmup 1 > $OUT/c5_seq$pol.json 2> $OUT/c5_seq$pol.err
  python3 -c "
import json
d=json.load(open('$OUT/c5_seq$pol.json')); print('config5 seq=$pol', d['value'], d['ms_per_step'], json.dumps(d['config5_mul_ctpt_u32']))"
done
