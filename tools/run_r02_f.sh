#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02f
mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_paillier.py tests/test_gpu_modexp.py tests/test_gpu_capi_sweep.py -x -q -m gpu 2>&1 | tail -3
for rr in 0 1; do for pol in fixed sliding; do
  PGPU_REGROWS=$rr PGPU_SECRET_EXP=$pol timeout 120 python bench.py --steps 10 --no-extras --no-cpu-baseline > $OUT/b.$rr.$pol.json 2> $OUT/b.$rr.$pol.err
  python - <<P
import json
try:
    d=json.load(open("$OUT/b.$rr.$pol.json"))
    print("regrows=$rr $pol", "value", d["value"], "dec_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("regrows=$rr $pol FAILED", e, open("$OUT/b.$rr.$pol.err").read()[-400:])
P
done; done
