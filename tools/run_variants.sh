#!/bin/bash
# A/B of kernel variants (tools/build_variant.py): parity smoke + bench line per library
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/variants
mkdir -p $OUT
cd $REPO
for v in "$@"; do
  lib=$REPO/pailliercryptolib_amd/libpgpu_$v.so
  [ "$v" = "base" ] && lib=$REPO/pailliercryptolib_amd/libpgpu.so
  PGPU_LIB=$lib python -m pytest tests/test_gpu_paillier.py tests/test_gpu_modexp.py -x -q -m gpu 2>&1 | tail -2 > $OUT/$v.pytest
  for pol in fixed sliding; do
    PGPU_LIB=$lib PGPU_SECRET_EXP=$pol python bench.py --steps 10 --no-extras --no-cpu-baseline > $OUT/$v.$pol.json 2> $OUT/$v.$pol.err
    python - <<P
import json
try:
    d=json.load(open("$OUT/$v.$pol.json"))
    print("$v $pol", "value", d["value"], "dec_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "enc_ms", list(d["roofline"]["other_kernels"].values())[1]["ms"])
except Exception as e:
    print("$v $pol FAILED", e, open("$OUT/$v.$pol.err").read()[-500:])
P
  done
  cat $OUT/$v.pytest
done
