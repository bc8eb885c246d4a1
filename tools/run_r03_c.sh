#!/bin/bash
# r03: code-placement A/B of the decrypt kernel (PGPU_PHASE_PAD variants), API stage probe
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03c
mkdir -p $OUT
cd $REPO
for v in base pad1 pad2 pad3; do
  lib=$REPO/pailliercryptolib_amd/libpgpu_$v.so; [ $v = base ] && lib=$REPO/pailliercryptolib_amd/libpgpu.so
  for packed in 0 1; do
    for rows in 1 0; do
      PGPU_LIB=$lib PGPU_PACKED_DECRYPT=$packed PGPU_PAIR_ROWS=$rows timeout 120 python3 bench.py --no-extras --no-cpu-baseline --steps 10 > $OUT/b_${v}_p${packed}_r${rows}.json 2> $OUT/b_${v}_p${packed}_r${rows}.err
      python3 - <<PY
import json
try:
    d=json.load(open("$OUT/b_${v}_p${packed}_r${rows}.json")); print("$v packed=$packed rows=$rows", d["ms_per_step"], d["roofline"]["kernel_ms"], [ (k[:24],x["ms"]) for k,x in d["roofline"]["other_kernels"].items()])
except Exception as e: print("$v packed=$packed rows=$rows FAILED", e)
PY
    done
  done
done
g++ -O2 -std=c++17 -fopenmp -Iinclude -Itests/cpp -Ipailliercryptolib_amd/csrc/host tools/api_probe.cpp -Lpailliercryptolib_amd -lipcl_amd -lpgpu -Wl,-rpath,$REPO/pailliercryptolib_amd -o /tmp/api_probe 2> $OUT/probe_build.err
python3 -c "from pailliercryptolib_amd import build; build.build_ipcl()" > /dev/null 2>&1
timeout 200 /tmp/api_probe > $OUT/api_probe.txt 2>&1; echo "probe rc=$?"
cat $OUT/api_probe.txt
