#!/bin/bash
# round 4, lease E: one-lane decrypt kernel -- parity, and cycles / clock against the sequential-halves kernel (1024-bit keys)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04e
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "lane_decrypt" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for pol in 1 0; do
  PGPU_LANE_DECRYPT=$pol timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pol$pol -- python $REPO/tools/bench_keysizes.py 65536 > $OUT/trace_pol$pol.log 2>&1
  PGPU_LANE_DECRYPT=$pol timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_pol$pol -- python $REPO/tools/bench_keysizes.py 65536 > $OUT/pmc_pol$pol.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04e"
for pol in (1, 0):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{out}/pmc_pol{pol}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void pgpu::", "")
            if ("lane" in k or "<2, 10>" in k) and int(r["Grid_Size"]) > 100000:
                agg[k + " vgpr " + r["VGPR_Count"] + "+" + r["Accum_VGPR_Count"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        c = {a: sum(v) / len(v) for a, v in cs.items()}
        print("PGPU_LANE_DECRYPT", pol, k, {a: round(b) for a, b in c.items()}, "| cycles per XCD", round(c["GRBM_GUI_ACTIVE"] / 8),
              "VALU instructions per SIMD", round(c["SQ_INSTS_VALU"] / 1024), "cycles per instruction",
              round(c["GRBM_GUI_ACTIVE"] / 8 / (c["SQ_INSTS_VALU"] / 1024), 3))
    for fn in glob.glob(f"{out}/trace_pol{pol}/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(fn)):
            if "lane" in r["Name"] or "<2, 10>" in r["Name"]:
                print("PGPU_LANE_DECRYPT", pol, r["Name"][:60], "calls", r["Calls"], "avg ms", float(r["AverageNs"]) / 1e6, "min", float(r["MinNs"]) / 1e6)
PY
