#!/bin/bash
# sequential-halves decrypt: lower threshold check, and the headline bench with two batches in flight when both lanes'
# decrypts take the sequential form (512 wavefronts each)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03o
mkdir -p $OUT
cd $REPO
timeout 600 python3 -m pytest tests/test_gpu_pair_rows.py -m gpu -x -q > $OUT/pytest_pair.log 2>&1; echo "pytest pair rc=$?"; tail -3 $OUT/pytest_pair.log
echo "decrypt sizes default"; timeout 300 python3 tools/bench_decrypt_sizes.py 8192 16384
for pol in 1 2 1 2; do
  PGPU_SEQ_DECRYPT=$pol timeout 300 python3 bench.py --steps 40 --warmup 4 > $OUT/bench_seq$pol.json 2> $OUT/bench_seq$pol.err
  python3 -c "
import json
d=json.load(open('$OUT/bench_seq$pol.json')); print('bench seq=$pol', d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline']['kernel_ms'])"
done
timeout 300 python3 bench.py --config 4 --steps 3 --warmup 1 > $OUT/c4.json 2> $OUT/c4.err; python3 -c "
import json
d=json.load(open('$OUT/c4.json')); print('config4', d['value'], d['ms_per_step'], d['roofline'])"
