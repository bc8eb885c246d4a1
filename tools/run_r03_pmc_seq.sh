#!/bin/bash
# PMC evidence for the sequential-halves forms: VALU instructions per launch, paired vs sequential kernels (config 4; decrypt of
# 16384 ciphertexts under a 2048-bit key).  Counters in their own passes, no trace domains.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r03seq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pol in 0 1; do
  PGPU_SEQ_DECRYPT=$pol timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/c4_pol$pol -- python $REPO/bench.py --config 4 --steps 2 --warmup 1 > $OUT/c4_pol$pol.log 2>&1
  PGPU_SEQ_DECRYPT=$pol timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/d16k_pol$pol -- python $REPO/tools/bench_decrypt_sizes.py 16384 > $OUT/d16k_pol$pol.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r03seq"
for d in sorted(glob.glob(out + "/*_pol*")):
    if not os.path.isdir(d): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void pgpu::", "")
            if int(r["Grid_Size"]) < 64 * 1000: continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "decrypt" in k or "encrypt" in k:
            print(os.path.basename(d), k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(cs["SQ_WAVES"]))
PY
