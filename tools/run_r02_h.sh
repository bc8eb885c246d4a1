#!/bin/bash
# A/B of the CRT-decrypt exponentiation: split form (hensel.hpp) on / off, both exponent policies.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02h
mkdir -p $OUT
cd $REPO
for hh in 1 0; do for pol in fixed sliding; do
  PGPU_HENSEL=$hh PGPU_SECRET_EXP=$pol timeout 120 python bench.py --steps 10 --no-extras --no-cpu-baseline > $OUT/b.$hh.$pol.json 2> $OUT/b.$hh.$pol.err
  python - <<P
import json
try:
    d=json.load(open("$OUT/b.$hh.$pol.json"))
    print("hensel=$hh $pol", "value", d["value"], "dec_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("hensel=$hh $pol FAILED", e, open("$OUT/b.$hh.$pol.err").read()[-400:])
P
done; done
