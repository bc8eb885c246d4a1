#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PGPU_AB_DECRYPT=1
CMD="python $REPO/bench.py --in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("pmc_sq","pmc_sq2"):
    agg=collections.defaultdict(list)
    for fn in glob.glob("$OUT/%s/*/*_counter_collection.csv"%d):
        for r in csv.DictReader(open(fn)):
            if "hensel_decrypt" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print(k, sum(v)/len(v))
PY
grep -h "hensel" $OUT/trace/*/*_kernel_stats.csv
