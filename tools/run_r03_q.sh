#!/bin/bash
# two batches in flight, both decrypts in the sequential-halves form (512 wavefronts each), workgroups claiming more than half
# a CU's LDS so that the two launches spread over all CUs
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03q
mkdir -p $OUT
cd $REPO
for cfg in "1 0" "2 0" "2 84000" "2 100000" "1 0" "2 84000"; do
  set -- $cfg
  PGPU_SEQ_DECRYPT=$1 PGPU_SEQ_LDS_PAD=$2 timeout 300 python3 bench.py --steps 40 --warmup 4 > $OUT/bench_$1_$2.json 2> $OUT/bench_$1_$2.err
  python3 -c "
import json
d=json.load(open('$OUT/bench_$1_$2.json')); print('bench seq=$1 pad=$2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
