#!/bin/bash
# round 4, lease A: GPU suite on the slimmed build; batches in flight on 1..4 lanes under each policy; decrypt size sweep
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04a
mkdir -p $OUT
cd $REPO
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for pol in 4 1 0; do
  timeout 300 python tools/probe_lanes.py --policy $pol > $OUT/lanes_pol$pol.log 2>&1
  cat $OUT/lanes_pol$pol.log
done
for pol in 2 0; do
  PGPU_SEQ_DECRYPT=$pol timeout 300 python tools/bench_decrypt_sizes.py 8192 16384 32768 65536 > $OUT/sizes_pol$pol.log 2>&1
  cat $OUT/sizes_pol$pol.log
done
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
