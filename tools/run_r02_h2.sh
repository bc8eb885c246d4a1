#!/bin/bash
# split form on / off: headline, small batches at the ipcl:: API, config 4
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02h2
mkdir -p $OUT
cd $REPO
for hh in 1 0; do
  PGPU_HENSEL=$hh timeout 120 python bench.py --steps 10 --no-extras --no-cpu-baseline > $OUT/b.$hh.json 2> $OUT/b.$hh.err
  python - <<P
import json
try:
    d=json.load(open("$OUT/b.$hh.json"))
    print("hensel=$hh", "value", d["value"], "dec_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("hensel=$hh FAILED", e, open("$OUT/b.$hh.err").read()[-400:])
P
  PGPU_HENSEL=$hh timeout 300 ./pailliercryptolib_amd/ipcl_api_bench > $OUT/api.$hh.txt 2>&1
  grep -i "decrypt" $OUT/api.$hh.txt | head -12
  PGPU_HENSEL=$hh timeout 300 python bench.py --config 4 --steps 3 --warmup 1 > $OUT/c4.$hh.json 2> $OUT/c4.$hh.err
  python - <<P
import json
try:
    d=json.load(open("$OUT/c4.$hh.json"))
    print("config4 hensel=$hh", "value", d["value"], "ms", d["ms_per_step"])
except Exception as e:
    print("config4 hensel=$hh FAILED", e, open("$OUT/c4.$hh.err").read()[-400:])
P
done
