"""Copies the round-3 evidence from gpurun_out/ into profiles/r03_* (tools/, bookkeeping only)."""
import json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)


def bench_line(src, dst):
    txt = open(src).read().strip().splitlines()
    j = json.loads([l for l in txt if l.startswith("{")][-1])
    open(dst, "w").write(json.dumps(j) + "\n")
    return j


pairs = [("gpurun_out/r03e/bench_n1.json", "profiles/r03_bench_n1.json"),
         ("gpurun_out/r03e/bench_f1.json", "profiles/r03_bench_n1_one_in_flight.json"),
         ("gpurun_out/r03d/bench_c5.json", "profiles/r03_bench_config5_n1.json"),
         ("gpurun_out/r03b/bench_c5_words.json", "profiles/r03_bench_config5_n1_montgomery_words.json"),
         ("gpurun_out/r03d/bench_c4.json", "profiles/r03_bench_config4_n1.json")]
for s, d in pairs:
    if os.path.exists(s):
        j = bench_line(s, d)
        print(d, j["value"], j["ms_per_step"], j["roofline"].get("kernel_ms"), j["roofline"].get("frac"))
for s, d in [("gpurun_out/r03f/ipcl_api_bench.txt", "profiles/r03_ipcl_api_bench.txt"),
             ("gpurun_out/r03f/api_probe.txt", "profiles/r03_api_stage_probe.txt"),
             ("gpurun_out/r03a/ubench_phase.txt", "profiles/r03_ubench_phase.txt"),
             ("gpurun_out/r03a/host_glue.txt", "profiles/r03_host_glue_threads.txt")]:
    if os.path.exists(s):
        shutil.copy(s, d)
# code placement A/B (tools/run_r03_c.sh): decrypt-kernel ms per variant
rows = []
for v in ("base", "pad1", "pad2", "pad3"):
    for packed in (0, 1):
        f = f"gpurun_out/r03c/b_{v}_p{packed}_r1.json"
        if os.path.exists(f):
            j = json.load(open(f))
            rows.append((v, packed, j["roofline"]["kernel_ms"], j["ms_per_step"]))
if rows:
    with open("profiles/r03_code_placement.txt", "w") as f:
        f.write("# hensel_decrypt_kernel<2,19> on the bench's decrypt launch (8192 ciphertexts, one wavefront per SIMD), same source,\n"
                "# main loop shifted by PGPU_PHASE_PAD: bit 0 = one 4-byte s_nop in front of it, bit 1 = .p2align 6 in front of it.\n"
                "# build 0 = full register budget (291 VGPRs), build 1 = the 256-register build.  tools/run_r03_c.sh, one MI355X box.\n"
                "# In the fast variants 88 % of the 8-byte instructions of the squaring loop start at 0 mod 8, in the slow ones 12 %\n"
                "# (llvm-objdump of the code objects): an 8-byte instruction that starts at 4 mod 8 costs a lone wavefront ~20 % more.\n"
                "variant  build  decrypt_kernel_ms  ms_per_step\n")
        for r in rows:
            f.write("%-8s %-6d %-18.3f %.3f\n" % r)
    print(open("profiles/r03_code_placement.txt").read())
