"""gpurun_out/r06_place_pad.txt + r06_threads_small.txt -> the tables under profiles/ (tools/run_place_pad.sh, run_threads_small.sh)"""
import json, re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def table(path, key):
    tag, T = None, {}
    for l in open(path):
        l = l.strip()
        if l.startswith(key + "="):
            tag = l.split("=")[1]
            continue
        try:
            d = json.loads(l)
        except ValueError:
            continue
        m = re.search(r"(batch|vector of) (\d+)", d["what"])
        k = "us_per_encrypt_plus_decrypt" if "us_per_encrypt_plus_decrypt" in d else "us_per_mul"
        op = "encrypt + decrypt" if k.startswith("us_per_enc") else "CipherText * PlainText"
        T.setdefault((op, int(m.group(2)), d["threads"]), {}).setdefault(tag, []).append(d[k])
    return T

def fmt(v):
    return " / ".join(f"{x:7.1f}" for x in v)

def main():
    out = ["Placement pad (launch.hpp: place_pad), MI355X, 2048-bit key, ipcl:: API from T host threads (tools/run_place_pad.sh)",
           "  us per call AGGREGATE over the threads (wall time / (rounds x threads)), two runs each; default policy otherwise",
           "",
           f"{'op':24s} {'batch':>6s} {'threads':>7s}   {'PGPU_PLACE_PAD=0 (round 5)':>28s}   {'PGPU_PLACE_PAD=192 (default)':>28s}"]
    for (op, n, t), v in table(os.path.join(ROOT, "gpurun_out", "r06_place_pad.txt"), "PGPU_PLACE_PAD").items():
        out.append(f"{op:24s} {n:6d} {t:7d}   {fmt(v.get('0', [])):>28s}   {fmt(v.get('192', [])):>28s}")
    wp = os.path.join(ROOT, "gpurun_out", "r06_wave_pad.txt")
    if os.path.exists(wp):
        out += ["", "The same with the pad also on the latency forms' kernels (hensel_decrypt_wave_kernel: its window table alone lets two",
                "workgroups share a CU; hensel_fb_encrypt_wave_kernel: no LDS of its own) -- tools/run_wave_pad.sh:", ""]
        for (op, n, t), v in table(wp, "PGPU_PLACE_PAD").items():
            out.append(f"{op:24s} {n:6d} {t:7d}   {fmt(v.get('0', [])):>28s}   {fmt(v.get('192', [])):>28s}")
    rr = os.path.join(ROOT, "gpurun_out", "r06_rr_adapt.txt")
    if os.path.exists(rr):
        out += ["", "Four API threads: the part-chip forms of the adaptive policy (PGPU_RR_ADAPT=3) against the lone caller's forms",
                "(PGPU_RR_ADAPT=0), measured with the part-chip forms entered from 1024 elements per launch (tools/run_rr_adapt.sh; below",
                "that both columns run the same forms -- before the threshold the left column read 1033 us at 64 and 1045 us at 256).",
                "The threshold that ships is 4096 (policy.hpp: kRrAdaptMinCount).", "",
                f"{'op':24s} {'batch':>6s} {'threads':>7s}   {'PGPU_RR_ADAPT=3 (default)':>28s}   {'PGPU_RR_ADAPT=0':>28s}"]
        for (op, n, t), v in table(rr, "PGPU_RR_ADAPT").items():
            out.append(f"{op:24s} {n:6d} {t:7d}   {fmt(v.get('3', [])):>28s}   {fmt(v.get('0', [])):>28s}")
    out += ["",
            "All tables above are from the build with the two fixes this investigation led to: (1) BaseText::ensureHost downloads",
            "outside its address-slot lock -- before, four threads were bimodal from run to run (CT x PT 1024 x 4: 1.45-1.5 ms in some",
            "runs, 1.9-2.8 ms in others: one thread's operator* starved 60-350 ms behind another thread's download lock; all 16 of 16",
            "runs 1.45 ms after the fix, tools/run_stall.sh); (2) workspaces grow through the block arena, not hipMallocAsync",
            "(profiles/r06_thread_race.txt: a wrong-result race of four threads' first decrypts)."]
    open(os.path.join(ROOT, "profiles", "r06_place_pad.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))
    out = ["Small batches from several host threads through the ipcl:: API (2048-bit key), MI355X (tools/run_threads_small.sh)",
           "  us per call AGGREGATE over the threads; PGPU_WAVE_FORMS=1 is the default policy (latency forms for a caller that finds",
           "  SIMDs to spare: wavefronts x (1 + active neighbour lanes) <= 1024), PGPU_WAVE_FORMS=0 the throughput forms only",
           "",
           f"{'op':24s} {'batch':>6s} {'threads':>7s}   {'latency forms (default)':>24s}   {'throughput forms only':>24s}"]
    for (op, n, t), v in table(os.path.join(ROOT, "gpurun_out", "r06_threads_small.txt"), "PGPU_WAVE_FORMS").items():
        out.append(f"{op:24s} {n:6d} {t:7d}   {fmt(v.get('1', [])):>24s}   {fmt(v.get('0', [])):>24s}")
    open(os.path.join(ROOT, "profiles", "r06_ipcl_api_threads_small.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))

if __name__ == "__main__":
    main()
