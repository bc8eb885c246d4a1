#!/bin/bash
# sequential-halves decrypt kernel: parity, then config 4 and large 2048-bit batches with it on / off
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03n
mkdir -p $OUT
cd $REPO
timeout 600 python3 -m pytest tests/test_gpu_pair_rows.py -m gpu -x -q -k "sequential" > $OUT/pytest_seq.log 2>&1; echo "pytest seq rc=$?"; tail -4 $OUT/pytest_seq.log
for pol in 0 1; do
  PGPU_SEQ_DECRYPT=$pol timeout 300 python3 bench.py --config 4 --steps 3 --warmup 1 > $OUT/c4_seq$pol.json 2> $OUT/c4_seq$pol.err
  python3 -c "
import json
d=json.load(open('$OUT/c4_seq$pol.json')); print('config4 seq=$pol', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  echo "decrypt sizes seq=$pol"; PGPU_SEQ_DECRYPT=$pol timeout 300 python3 tools/bench_decrypt_sizes.py 16384 32768 65536
done
PGPU_SEQ_DECRYPT=2 timeout 200 python3 tools/bench_decrypt_sizes.py 8192 16384
