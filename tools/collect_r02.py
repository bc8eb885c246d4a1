"""Copies the outputs of tools/run_r02_final.sh (gpurun_out/r02final, gpurun_out/prof_r02) into profiles/r02_* and
prints the numbers DESIGN.md section 4 quotes (tools/, bookkeeping only)."""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
subprocess.run([sys.executable, "tools/summarize_prof.py", "r02"], check=True, stdout=subprocess.DEVNULL)
names = ["bench_n1", "bench_n1_fullwidth_decrypt", "bench_n2_pool_1dev", "bench_n2_torchrun_1dev", "bench_config4_n1",
         "bench_config5_n1", "bench_config5_n2_pool_1dev"]
for f in names:
    txt = open(f"gpurun_out/r02final/{f}.json").read().strip().splitlines()
    j = json.loads([l for l in txt if l.startswith("{")][-1])
    open(f"profiles/r02_{f}.json", "w").write(json.dumps(j) + "\n")
    r = j["roofline"]
    print(f, j["value"], j["ms_per_step"], r.get("kernel", "")[:36], r.get("kernel_ms"), r.get("frac"), r.get("executed_frac"),
          {k[:30]: v.get("ms") for k, v in r.get("other_kernels", {}).items()})
    for k in ("config5_mul_ctpt_u32", "end_to_end"):
        if k in j and f.startswith("bench_config"):
            print("   ", k, j[k])
for src, dst in (("ipcl_api_bench.txt", "r02_ipcl_api_bench.txt"),
                 ("ipcl_api_bench_fullwidth_decrypt.txt", "r02_ipcl_api_bench_fullwidth_decrypt.txt"),
                 ("small_batch_cpu.txt", "r02_small_batch_cpu_ifma.txt")):
    shutil.copy(f"gpurun_out/r02final/{src}", f"profiles/{dst}")
open("profiles/r02_rocprofv3_trace_tail.log", "w").write("".join(open("gpurun_out/prof_r02/trace.log").readlines()[-3:]))
j = json.load(open("profiles/r02_bench_n1.json"))
for k in ("decrypt_only_modexps_per_s", "end_to_end", "api_level", "config2_nondjn", "sliding_window_policy"):
    print(k, j[k])
cb = j["cpu_baseline"]
print({k: v for k, v in cb.items() if k not in ("legs", "sample")}, {k: (v["value"], v["one_thread"]["value"]) for k, v in cb["legs"].items()})
print(open("profiles/r02_rocprofv3_kernel_trace_fullbatch.csv").read())
d = json.load(open("profiles/r02_pmc_counters.json"))
for k, v in d.items():
    c = v["counters_avg_per_dispatch"]
    print(k, "scratch", v["dispatch"]["Scratch_Size"], "valu/wave", round(v.get("valu_insts_per_wave", 0)), "hbm MB",
          round(v.get("hbm_bytes_fetch_x2_corrected", 0) / 1e6), "lds insts", c.get("SQ_INSTS_LDS"), "bank conflicts",
          c.get("SQ_LDS_BANK_CONFLICT"), "of", c.get("SQ_LDS_IDX_ACTIVE"), "wave cycles/4", c.get("SQ_WAVE_CYCLES"))
