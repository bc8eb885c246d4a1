#!/bin/bash
# A/B of the pair-form filler parameters (PGPU_PAIR_FA/FB/FC) on the aligned 256-register decrypt kernel, one and two batches in flight
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03g
mkdir -p $OUT
cd $REPO
for v in base f002 f233 f442 f221 base; do
  lib=$REPO/pailliercryptolib_amd/libpgpu_$v.so; [ $v = base ] && lib=$REPO/pailliercryptolib_amd/libpgpu.so
  for fl in 1 2; do
    PGPU_LIB=$lib timeout 120 python3 bench.py --in-flight $fl --no-extras --no-cpu-baseline --steps 20 > $OUT/b_${v}_f${fl}.json 2> $OUT/b_${v}_f${fl}.err
    python3 -c "
import json
d=json.load(open('$OUT/b_${v}_f${fl}.json')); print('$v in-flight $fl', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
