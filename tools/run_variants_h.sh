#!/bin/bash
# A/B of libpgpu variants (tools/build_variant.py) on the bench's decrypt launch: run_variants_h.sh <name>...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for v in "$@"; do
  lib=$REPO/pailliercryptolib_amd/libpgpu_$v.so
  [ "$v" = base ] && lib=$REPO/pailliercryptolib_amd/libpgpu.so
  for rep in 1 2; do
  PGPU_LIB=$lib timeout 120 python bench.py --steps 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
