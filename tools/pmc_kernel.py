"""Average the counters of one kernel out of a rocprofv3 --pmc output directory.  usage: pmc_kernel.py <dir> <name-substring>"""
import collections, csv, glob, sys
agg, meta = collections.defaultdict(list), None
for fn in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["Kernel_Name"][:70], "vgpr", r["VGPR_Count"], "lds", r["LDS_Block_Size"], "grid", r["Grid_Size"])
print(meta)
print({k: sum(v) / len(v) for k, v in agg.items()})
