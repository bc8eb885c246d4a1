"""Copies the round-6 evidence from gpurun_out/ into profiles/r06_* and refreshes profiles/pmc_summary.json
(tools/, bookkeeping only).  Inputs: tools/profile_r06.sh (gpurun_out/prof_r06{f1,f2,c4,c5}, gpurun_out/r06p)."""
import collections, csv, glob, json, os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
KEEP = ("modexp_kernel", "hensel_", "crt_kernel", "modmul_kernel", "fb_", "pair_ops_kernel", "pair_mul_seq_kernel")


def short(name):
    return name.split("(")[0].replace("void pgpu::", "") if any(k in name for k in KEEP) else None


def bench_line(src, dst):
    """src: the DETAIL record of a bench.py run (bench_detail.json as tools/profile_r06.sh copied it); the compact contract
    line of the same run lies beside it as *.line.json and is copied beside the record"""
    j = json.load(open(src))
    open(dst, "w").write(json.dumps(j) + "\n")
    ln = src[:-5] + ".line.json"
    if os.path.exists(ln):
        txt = [l for l in open(ln).read().strip().splitlines() if l.startswith("{")]
        if txt:
            open(dst[:-5] + ".line.json", "w").write(txt[-1] + "\n")
    return j


build = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
pmc_all = {}
for tag, label in (("f1", ""), ("f2", "_two_in_flight"), ("f4", "_four_in_flight"), ("c4", ""), ("c5", "")):
    src = f"gpurun_out/prof_r06{tag}"
    if not os.path.isdir(src):
        continue
    pre = "profiles/r06" + ("c4" if tag == "c4" else "c5" if tag == "c5" else "")
    stats = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(f"{pre}_rocprofv3_kernel_stats{label}.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    tr = glob.glob(f"{src}/trace/*/*_kernel_trace.csv")
    if tr and tag in ("f1", "f2", "f4"):
        # the launches of the library's kernels with their start / end timestamps (ns, relative to the first): under
        # overlap (f2) this is what the measured fractions can be recomputed from -- union of intervals, concurrency
        rows = [r for r in csv.DictReader(open(tr[0])) if short(r["Kernel_Name"])]
        t0 = min(int(r["Start_Timestamp"]) for r in rows)
        with open(f"{pre}_rocprofv3_kernel_trace{label}.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "Queue_Id", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Start_ns", "End_ns", "Duration_ns"])
            for r in rows:
                s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
                w.writerow([short(r["Kernel_Name"]), r.get("Queue_Id", ""), r.get("Grid_Size", r.get("Grid_Size_X", "")),
                            r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("LDS_Block_Size", ""), s, e, e - s])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        for fn in glob.glob(f"{src}/{d}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(fn)):
                s = short(r["Kernel_Name"])
                if not s or int(r["Grid_Size"]) < 64 * 250:
                    continue
                agg[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[s] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    out = {}
    for s, cs in agg.items():
        c = {k: sum(v) / len(v) for k, v in cs.items()}
        out[s] = {"dispatch": meta[s], "launches_averaged": {k: len(v) for k, v in cs.items()}, "counters_avg_per_dispatch": c}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            # rocprofv3 reports KiB; gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide reads (MI355X_MICROARCH.md, HBM)
            out[s]["hbm_bytes_raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
            out[s]["hbm_bytes_fetch_x2_corrected"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
            out[s]["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / max(c["SQ_WAVES"], 1)
    if out:
        json.dump(out, open(f"{pre}_pmc_counters{label}.json", "w"), indent=1)
        pmc_all[tag] = out

old_summary = json.load(open("profiles/pmc_summary.json")) if os.path.exists("profiles/pmc_summary.json") else {}
summary = {"source": "profiles/r06*_pmc_counters*.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in passes of their own, tools/profile_r06.sh; "
                     "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 averaged over the full-size launches of the command, the x2 on "
                     "FETCH per MI355X_MICROARCH.md (HBM): gfx950 tallies wide reads at half their size)",
           "build": build, "collected": __import__("datetime").date.today().isoformat() + " (round 6, tools/profile_r06.sh)"}


def pick(tag, prefix):
    for k, v in pmc_all.get(tag, {}).items():
        if k.startswith(prefix) and "hbm_bytes_fetch_x2_corrected" in v:
            return k, v["hbm_bytes_fetch_x2_corrected"], v["hbm_bytes_raw"]
    return None, None, None


for key, tag, prefix in (("modexp_decrypt", "f1", "hensel_decrypt_kernel<"), ("seq_decrypt", "f2", "hensel_decrypt_seq_kernel<"),
                         ("ps_decrypt", "f4", "hensel_decrypt_ps_kernel<"), ("fb_encrypt_seq_quarter", "f4", "hensel_fb_encrypt_seq_kernel<"),
                         ("fb_encrypt", "f1", "hensel_fb_encrypt_kernel<"), ("fb_encrypt_seq", "f2", "hensel_fb_encrypt_seq_kernel<"),
                         ("config4_decrypt", "c4", "hensel_decrypt_seq_kernel<"), ("config4_decrypt", "c4", "hensel_decrypt_ps_kernel<"),
                         ("config4_encrypt", "c4", "hensel_fb_encrypt_seq_kernel<"),
                         ("ct_add", "c5", "pair_mul_seq_kernel<"), ("ct_mul", "c5", "hensel_modexp_seq_kernel<")):
    name, b, raw = pick(tag, prefix)
    if name:
        summary[key + "_kernel"] = name
        summary[key + "_hbm_bytes_per_launch"] = b
        summary[key + "_hbm_bytes_per_launch_raw"] = raw
if "ct_add_hbm_bytes_per_launch" in summary:
    summary["ct_add_pair_mul_hbm_bytes_per_launch"] = summary["ct_add_hbm_bytes_per_launch"]
if len(summary) > 3:
    # a partial run (RUNS=... tools/profile_r06.sh) refreshes its own keys only; "build" / "collected" then name the latest pass
    for k, v in old_summary.items():
        summary.setdefault(k, v)
    if old_summary.get("build") and old_summary.get("build") != build and len(pmc_all) < 5:
        summary["build_note"] = "keys of passes not re-run keep the values of build " + old_summary["build"]
    json.dump(summary, open("profiles/pmc_summary.json", "w"), indent=1)
    print(json.dumps(summary, indent=1))

pairs = [("gpurun_out/r06p/bench_n1.json", "profiles/r06_bench_n1.json"),
         ("gpurun_out/r06p/bench_f1.json", "profiles/r06_bench_n1_one_in_flight.json"),
         ("gpurun_out/r06p/bench_f2.json", "profiles/r06_bench_n1_two_in_flight.json"),
         ("gpurun_out/r06p/bench_c4.json", "profiles/r06_bench_config4_n1.json"),
         ("gpurun_out/r06p/bench_c5.json", "profiles/r06_bench_config5_n1.json"),
         ("gpurun_out/r06p/bench_n8_pool_1dev.json", "profiles/r06_bench_n8_pool_1dev.json"),
         ("gpurun_out/r06p/bench_c4_n8_pool_1dev.json", "profiles/r06_bench_config4_n8_pool_1dev.json"),
         ("gpurun_out/r06p/bench_c5_n8_pool_1dev.json", "profiles/r06_bench_config5_n8_pool_1dev.json")]
for s, d in pairs:
    if os.path.exists(s):
        try:
            j = bench_line(s, d)
            print(d, j["value"], j["ms_per_step"], j["roofline"].get("kernel_ms"), j["roofline"].get("frac"), j.get("host_issue_ms_per_step"))
        except Exception as e:                                  # noqa: BLE001
            print(d, "unreadable:", e)
for s, d in [("gpurun_out/r06p/ipcl_api_bench.txt", "profiles/r06_ipcl_api_bench.txt"),
             ("gpurun_out/r06p/small_batch_cpu_ifma.txt", "profiles/r06_small_batch_cpu_ifma.txt"),
             ("gpurun_out/r06p/wave_form_sizes.txt", "profiles/r06_wave_form_sizes.txt"),
             ("gpurun_out/r06p/lanes.txt", "profiles/r06_lanes.txt"),
             ("gpurun_out/r06p/trace_2lanes.txt", "profiles/r06_trace_2lanes.txt"),
             ("gpurun_out/r06p/trace_4lanes.txt", "profiles/r06_trace_4lanes.txt"),
             ("gpurun_out/r06p/lanes_masked.txt", "profiles/r06_lanes_masked_gather.txt"),
             ("gpurun_out/r06p/big_ps1.txt", "profiles/r06_decrypt_65536_ps.txt"),
             ("gpurun_out/r06p/big_ps0.txt", "profiles/r06_decrypt_65536_seq.txt")]:
    if os.path.exists(s):
        txt = [l for l in open(s).read().splitlines() if "amdgpu.ids" not in l]
        open(d, "w").write("\n".join(txt) + "\n")
if os.path.exists("gpurun_out/r06p/two_callers.txt"):
    with open("profiles/r06_two_callers.txt", "w") as f:
        f.write("# tools/probe_two_callers.py <pageable|pinned> <callers> <rounds>: host threads calling pgpu_paillier_encrypt +\n"
                "# pgpu_paillier_decrypt_crt (8192 x 2048-bit) synchronously on host arrays of their own; per-caller start / encrypt / decrypt\n"
                "# wall times (ms) and the aggregate rate.  PGPU_D2H_PRESYNC=1 (the default): a download is handed to the copy engine only\n"
                "# once the kernels in front of it have run -- otherwise it parks the engine's ring on that kernel and the other caller's\n"
                "# uploads wait behind it (=0: the behaviour before).  PGPU_HOST_ADAPT=1: the callers' launches take the half-chip forms of\n"
                "# the adaptive policy (off by default: slower for synchronous callers).\n")
        f.write("".join(l for l in open("gpurun_out/r06p/two_callers.txt") if not l.startswith("+")))
if os.path.exists("gpurun_out/r06p/ipcl_api_threads.txt"):
    with open("profiles/r06_ipcl_api_threads.txt", "w") as f:
        f.write("# pailliercryptolib_amd/ipcl_api_bench --threads T 8192 8 (tests/cpp/ipcl_bench.cpp): T host threads, each\n"
                "# ipcl::PublicKey::encrypt + PrivateKey::decrypt with vector<BigNumber> in and out (the benchmark key: 2047-bit injected r)\n")
        f.write("".join(l for l in open("gpurun_out/r06p/ipcl_api_threads.txt") if not l.startswith("+")))
if os.path.exists("gpurun_out/r06p/ipcl_api_threads_small.txt"):
    shutil.copy("gpurun_out/r06p/ipcl_api_threads_small.txt", "profiles/r06_ipcl_api_threads_small.txt")
ks = [f"gpurun_out/r06p/keysizes_{c}.txt" for c in (16384, 65536)]
if all(os.path.exists(f) for f in ks):
    with open("profiles/r06_keysizes_split_on_off.txt", "w") as f:
        f.write("# tools/bench_keysizes.py <count> (tools/profile_r06.sh): resident batches per key class, wall time of the second call incl.\n"
                "# launch overhead, PGPU_HENSEL off / on; decrypt leg as a fraction of the int-ALU peak (39.32 T MAC32/s): executed by the\n"
                "# kernel that ran / useful count of the split form.  Build without the 4096-bit split forms (PGPU_BUILD_4096=0, the default).\n")
        for fn in ks:
            f.write("".join(l for l in open(fn) if "amdgpu.ids" not in l))
            f.write("\n")
    print(open("profiles/r06_keysizes_split_on_off.txt").read())
