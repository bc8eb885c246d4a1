#!/bin/bash
# sequential-halves CT + CT kernel: parity, then config 5 with it on / off, full-size hash of config 5
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03t
mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_pair_rows.py tests/test_gpu_sha256_fullsize.py -m gpu -x -q -k "sequential or config5 or config_5 or full" > $OUT/pytest_seq.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_seq.log
for pol in 0 1 0 1; do
  PGPU_SEQ_DECRYPT=$pol timeout 600 python3 bench.py --config 5 --steps 5 --warmup 1 > $OUT/c5_seq$pol.json 2> $OUT/c5_seq$pol.err
  python3 -c "
import json
d=json.load(open('$OUT/c5_seq$pol.json')); r=d['roofline']; print('config5 seq=$pol', d['value'], d['ms_per_step'], r['kernel'][:40], r['kernel_ms'], r['frac'], r['hbm_frac'])"
done
