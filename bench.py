#!/usr/bin/env python3
"""bench.py -- headline benchmark: 2048-bit Paillier modexps/sec (encrypt + CRT decrypt).

One "step" = one pass of the hot path over one synthetic batch that is already resident in HBM:
   encrypt  8192 plaintexts under the reference's fixed 2048-bit ISO/IEC 18033-6 key, DJN scheme
            (c = hs^r * (1+n*m) mod n^2: 8192 modexps, 4096-bit modulus, 1024-bit exponent)
   decrypt  those 8192 ciphertexts with CRT (16384 modexps, 2048-bit moduli p^2 / q^2,
            1024-bit exponents) incl. L-function and recombination
= 24576 modexps per step per GPU -- BASELINE.json configs[1] + configs[2], the configuration the
metric "2048-bit modexps/sec (encrypt+decrypt)" is quoted on; it mirrors the reference's BM_Encrypt /
BM_Decrypt (benchmark/bench_cryptography.cpp:73-121: same key, same HS_BN, DJN on).

How N GPUs are driven:
  * `python bench.py --gpus N` (plain): ONE process, the library's device pool (pgpu_init_all: one worker pair +
    streams per GPU, key images replicated by one RCCL broadcast over xGMI); the global batch of N x 8192
    elements is a sharded resident pgpu_batch, a step enqueues one launch per GPU.  This is also the N = 1 path.
  * under torch.distributed.run (WORLD_SIZE set): one process per GPU, backend nccl == RCCL; every rank
    runs the same resident step on its own GPU (a one-entry pool); the only collective is the broadcast
    of the key material from rank 0.
Both are weak scaling (8192 elements per GPU per step); timing = barrier/synchronise on both sides of exactly K
steps, MAX over ranks.  Consecutive steps rotate over the library's four batch lanes (--in-flight, default 4 since round 5:
four independent batches of 8192 in flight per GPU, each lane's launches on a quarter of the chip -- csrc/policy.cpp); every
step is a whole encrypt + decrypt of one batch, and every lane's results are checked after the timed region.  `--config 4|5` runs BASELINE.json configs[3] / configs[4] instead: a FIXED total batch
(65536 x 3072-bit encrypt+decrypt; 1 M x 2048-bit CT+CT and CT x PT) sharded over the N GPUs (strong scaling).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bench_line  # noqa: E402  (the compact contract line; the record built below goes to bench_detail.json)

BATCH = 8192
KEY_BITS = 2048
PEAK_TMAC32 = 39.32   # 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz: v_mad_u64_u32 is full rate (38.35 measured)
HBM_PEAK_GBS = 8000.0
K_MODEXP, K_MODMUL, K_CRT, K_FB = 1, 2, 3, 4


def algorithmic_mac32(mod_bits, exp_bits):
    """M(s) = 2s^2+s MAC32 per Montgomery multiplication over s = mod_bits/32 words;
    N(e) = e + ceil(e/w) + 2^w multiplications, w = 5 for e >= 128 else 2."""
    s = mod_bits // 32
    w = 5 if exp_bits >= 128 else 2
    return (2 * s * s + s) * (exp_bits + (exp_bits + w - 1) // w + (1 << w))


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def sig(x, digits=2):
    """roofline fractions are quoted with two significant digits: between boxes of the pool the same kernel differs
    by 2-4 % (rocprofv3 vs HIP events alone: ~4 %), so more digits would be noise"""
    if x is None or x == 0:
        return x
    from math import floor, log10
    return round(x, digits - 1 - int(floor(log10(abs(x)))))


def iso_key():
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "iso_kat.json")))
    return int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)


def synth(rank, count, nw, pw):
    rng = np.random.default_rng(1234 + rank)
    m = np.frombuffer(rng.bytes(count * nw * 8), dtype=np.uint64).reshape(count, nw).copy()
    m[:, -1] &= np.uint64((1 << 62) - 1)          # plaintexts < 2^(64 nw - 2) < n
    r = np.frombuffer(rng.bytes(count * pw * 8), dtype=np.uint64).reshape(count, pw).copy()   # full-width r
    return m, r


def collect_timing(L, cap):
    kinds = (ctypes.c_int * cap)()
    kms = (ctypes.c_double * cap)()
    n = L.pgpu_timing_collect(kinds, kms, cap)
    per = {}
    for i in range(n):
        per.setdefault(kinds[i], []).append(kms[i])
    return per


class Batches:
    """thin owner of pgpu_batch handles"""

    def __init__(self, L, check):
        self.L, self.check = L, check

    def up(self, arr):
        h = ctypes.c_void_p()
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        self.check(self.L.pgpu_batch_upload(ptr(arr), arr.shape[0], arr.shape[1], arr.shape[1], ctypes.byref(h)))
        return h

    def down(self, h):
        out = np.empty((self.L.pgpu_batch_count(h), self.L.pgpu_batch_words(h)), dtype=np.uint64)
        self.check(self.L.pgpu_batch_download(h, ptr(out)))
        return out

    def op(self, fn, *a):
        h = ctypes.c_void_p()
        self.check(fn(*a, ctypes.byref(h)))
        return h

    def free(self, *hs):
        for h in hs:
            if h:
                self.L.pgpu_batch_destroy(h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5],
                    help="2: BASELINE configs[1]+[2] (default, the headline); 4 / 5: configs[3] / configs[4]")
    ap.add_argument("--in-flight", type=int, default=4, choices=[1, 2, 3, 4],
                    help="resident batches in flight per GPU: consecutive steps rotate over that many of the library's "
                         "four batch lanes (default 4 since round 5: each lane then owns a QUARTER of the chip -- the decrypt "
                         "runs the one-lane product-scanning kernel, 256 wavefronts on 64 CUs, the encrypt two workgroups per "
                         "CU on the same quarter; 2: each lane owns half the chip with the sequential-halves kernels, the "
                         "round-4 headline: batches_in_flight_sweep in the line has every k)")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="length of the sustained-rate run behind the timed region (N = 1; 0 skips it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the N=1 side measurements (end-to-end, API level, non-DJN variant)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        if args.gpus != world:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        return run_ranks(args, world)
    return run_pool(args)


# ----------------------------------------------------------------------------------------------------------------
# plain invocation: one process, in-process device pool
# ----------------------------------------------------------------------------------------------------------------
def run_pool(args):
    if os.environ.get("BENCH_SINGLE_DEVICE") == "1":      # validation of N > 1 on a 1-GPU box: entries wrap around
        os.environ["PGPU_POOL_OVERSUBSCRIBE"] = "1"
    try:
        import torch                                      # one HIP runtime per process: let torch bring it up first
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        torch = None
    import pailliercryptolib_amd as pa
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import limbs_to_ints
    L = _capi.lib()
    N = args.gpus
    _capi.check(L.pgpu_init_all(N))
    pa.engine._initialized = True
    ctypes.CDLL(None).fflush(None)      # librccl prints a banner through C stdio: out before our one JSON line
    _capi.check(L.pgpu_set_min_shard(256))
    B = Batches(L, _capi.check)
    if args.config != 2:
        result = run_config45(args, pa, L, B, N)
        pa.terminate()
        ctypes.CDLL(None).fflush(None)
        bench_line.emit(result)                         # detail -> bench_detail.json + stderr; compact line last on stdout
        return
    p, q, hs = iso_key()
    n = p * q
    nw, pw = KEY_BITS // 64, KEY_BITS // 128
    pk = pa.PublicKey(n, KEY_BITS, hs=hs)
    sk = pa.PrivateKey(p, q)

    # ---- synthetic global batch (N x 8192), sharded over the pool, resident before the timed region ----
    parts = [synth(g, BATCH, nw, pw) for g in range(N)]
    m_host = np.concatenate([a for a, _ in parts])
    r_host = np.concatenate([b for _, b in parts])
    # one input set per batch lane (the same values): a step takes the set of its lane, its ciphertexts and plaintexts
    # inherit the lane, so consecutive steps run on different streams of every GPU and overlap
    nfl = args.in_flight
    sets = []
    for ln in range(nfl):
        _capi.check(L.pgpu_set_batch_lane(ln))
        sets.append((B.up(m_host), B.up(r_host)))
    _capi.check(L.pgpu_set_batch_lane(0))
    bm, br = sets[0]
    state = {"c": [None] * nfl, "out": [None] * nfl, "i": 0}

    def step():
        k = state["i"] % nfl
        state["i"] += 1
        B.free(state["c"][k], state["out"][k])
        state["c"][k] = B.op(L.pgpu_batch_encrypt, pk._h, sets[k][0], sets[k][1], 64 * pw)
        state["out"][k] = B.op(L.pgpu_batch_decrypt_crt, sk._h, state["c"][k])

    def sync_all():
        _capi.check(L.pgpu_synchronize())
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    # Priming (set-up, like the key's tables): two untimed steps on every batch lane, so that each lane has run the kernel
    # forms of the steady state once -- the adaptive policy picks a lane's form from what its neighbours are doing, and the
    # very first launches of a process see idle neighbours -- and owns its window-table workspace (a first launch in a new
    # form is a hipMallocAsync of ~170 MB: milliseconds of host time).  Then the W warm-up steps of the contract.
    priming = 2 * nfl if nfl > 1 else 0
    for _ in range(priming):
        step()
    sync_all()
    for _ in range(max(args.warmup, nfl)):
        step()
    sync_all()
    # live per-kernel timing: the library brackets every launch with HIP events on the launch stream (no
    # synchronisation inside the timed region) and records the kernel FORM it picked -- with several batches in flight
    # the adaptive policy chooses per launch, from what the neighbour lanes are doing (include/pgpu.h), so the line says
    # what actually ran.  With batches in flight the launches of the lanes overlap, so an event pair spans a SHARED
    # stretch of the chip: per-kernel times of a lone launch come from a short single-lane pass behind the timed region,
    # and the decrypt leg under overlap from a decrypt-only run on the same lanes (measured, not derived).
    _capi.check(L.pgpu_set_timing(1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = time.perf_counter() - t0          # the calling thread has queued every launch of the timed region
    sync_all()
    elapsed = time.perf_counter() - t0
    timed_trace = collect_trace(L, 8 * args.steps + 64)
    timed_forms = [(k, f, ms) for k, f, ms, _, _ in timed_trace]
    single_ms = None
    if nfl == 1:
        per_kind = forms_to_kinds(timed_forms)
    else:
        time.sleep(0.15)            # (the other lanes count as active for 50 ms after they were last fed: let that lapse)
        for _ in range(2):
            state["i"] = 0
            step()
        sync_all()
        collect_forms(L, 64)
        t1 = time.perf_counter()
        for _ in range(10):
            state["i"] = 0          # lane 0 only
            step()
        sync_all()
        single_ms = (time.perf_counter() - t1) / 10 * 1e3
        per_kind = collect_timing(L, 64)
    _capi.check(L.pgpu_set_timing(0))
    measured = {}
    if N == 1:
        measured["decrypt_leg"] = decrypt_leg_in_flight(L, B, sk, state["c"], nfl, nw)
        if args.sustain_seconds > 0:
            state["i"] = 0
            k = max(nfl, int(args.sustain_seconds / (elapsed / args.steps)) // nfl * nfl)
            sync_all()
            t2 = time.perf_counter()
            for _ in range(k):
                step()
            sync_all()
            dt = time.perf_counter() - t2
            measured["sustained"] = {"steps": k, "seconds": round(dt, 3), "ms_per_step": round(dt / k * 1e3, 4),
                                     "modexps_per_s": round(3 * BATCH * N * k / dt, 1),
                                     "what": "the same steps back to back for about %.0f s behind the timed region (the boxes "
                                             "hold a lower clock over seconds than over the 0.1 s of a 20-step run)" % args.sustain_seconds}

        # the same step with constant-address table access (pgpu_set_table_gather_policy(1)): every window-table / fixed-base
        # table candidate is read and the wanted one selected, as the reference's mbx_exp_mb8 gathers (mod_exp.cpp:508-516)
        if not args.no_extras:
            gather_before = L.pgpu_get_table_gather_policy()
            try:
                _capi.check(L.pgpu_set_table_gather_policy(1))
                state["i"] = 0
                for _ in range(2 * nfl):
                    step()
                sync_all()
                kh = 6 * nfl
                t3 = time.perf_counter()
                for _ in range(kh):
                    step()
                sync_all()
                dth = time.perf_counter() - t3
                okh = all(bool(np.array_equal(B.down(o), m_host)) for o in state["out"])
                measured["hardened"] = {"steps": kh, "ms_per_step": round(dth / kh * 1e3, 4), "modexps_per_s": round(3 * BATCH * N * kh / dth, 1),
                                        "round_trip_ok": okh,
                                        "what": "the resident step of the headline with secret_table_access = masked: the decrypt window tables "
                                                "(3-bit windows under this policy: 8 entries, 348 products instead of 32 entries, 235) and the DJN "
                                                "fixed-base product (255 products over 16-entry windows instead of 78 indexed ones) read every "
                                                "candidate and select; same lanes, same batch"}
            except Exception as e:                          # noqa: BLE001
                measured["hardened"] = {"error": repr(e)[:300]}
            finally:
                _capi.check(L.pgpu_set_table_gather_policy(gather_before))
                state["i"] = 0
                for _ in range(nfl):       # (the checked results below are those of the default policy)
                    step()
                sync_all()

    # ---- correctness of what was timed: full-size round trip + oracle spot checks ----
    ok = all(bool(np.array_equal(B.down(o), m_host)) for o in state["out"])
    all_c, all_out = state["c"], state["out"]
    state["c"], state["out"] = all_c[0], all_out[0]
    from oracle import paillier_oracle as orc
    opk = orc.PublicKey(n, KEY_BITS)
    opk.set_djn(hs)
    c_all = B.down(state["c"])
    # oracle spot checks: first, middle and last row of EVERY GPU's shard (a GPU that holds a bad key image or runs a
    # bad kernel cannot hide behind the shards that are fine; the round trip above already covers every element)
    idx = sorted({g * BATCH + o for g in range(N) for o in (0, BATCH // 2 + g, BATCH - 1)})
    want = opk.encrypt(limbs_to_ints(m_host[idx]), limbs_to_ints(r_host[idx]))
    got = limbs_to_ints(c_all[idx])
    bad = [i for i, (x, y) in zip(idx, zip(got, want)) if x != y]
    if bad:
        raise SystemExit(f"bench: ciphertext rows {bad} (GPUs {sorted({i // BATCH for i in bad})}) differ from the oracle")
    if not ok:
        raise SystemExit("bench: GPU results differ from the oracle / round trip failed")
    per_gpu = per_gpu_report(L, N, per_kind)

    fb = fixed_base_info(L, pk)
    result = headline(args, N, elapsed, per_kind, nw, pw,
                      f"in-process device pool x{N} (batch sharded; key images: {L.pgpu_pool_transport().decode()})",
                      decrypt_kernel(sk, BATCH, nw, KEY_BITS),
                      encrypt_kernel(pk, BATCH, nw, KEY_BITS, fb["window"] or int(os.environ.get("PGPU_FB_WINDOW", "13"))), fb)
    result["config"]["secret_exponent_policy"] = ["fixed-window", "sliding"][L.pgpu_get_secret_exponent_policy()]
    result["config"]["batches_in_flight_per_gpu"] = nfl
    result["config"]["lane_priming_steps"] = priming
    result["config"]["resident_ciphertext_form"] = ("pair rows (%d limbs)" % L.pgpu_batch_row_limbs(state["c"])
                                                    if L.pgpu_batch_row_limbs(state["c"]) else "Montgomery-form words")
    r = result["roofline"]
    r["useful_mac32_per_launch"] = decrypt_useful_mac32(nw, KEY_BITS) * 2 * BATCH
    r["frac_useful"] = sig(r["useful_mac32_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12 / PEAK_TMAC32)
    r["useful_note"] = ("useful = the multiply-accumulates the split form needs (sequential-halves count: symmetric a*a, no "
                        "lane idling through another's products); executed - useful = slots the paired kernel spends on "
                        "products nobody needs")
    result["kernel_forms_in_timed_region"] = forms_summary(timed_forms)
    if N == 1 and len(timed_trace) <= 400:
        kn = {K_MODEXP: "dec", K_CRT: "crt", K_FB: "enc"}
        result["timed_region_trace"] = {
            "what": "every launch of the timed region on GPU 0: [batch lane, kernel, form bits, start ms, duration ms] from the "
                    "library's HIP events (pgpu_timing_collect_trace); start is relative to the first launch",
            "launches": [[ln, kn.get(k, k), f, round(st, 3), round(ms, 3)] for k, f, ms, ln, st in timed_trace]}
    result["host_issue_ms_per_step"] = round(t_issue / args.steps * 1e3, 4)
    result["host_issue_note"] = ("time the ONE calling thread spends queueing a step's launches over the %d pool entries (3 launches "
                                 "+ allocations per GPU and step); %.1f %% of a step" % (N, 100 * t_issue / elapsed))
    if nfl > 1:
        result["config"]["workload"] += ("; %d batches in flight per GPU: consecutive steps rotate over %d of the library's "
                                         "batch lanes (streams), exactly K steps timed" % (nfl, nfl))
        result["one_batch_in_flight"] = {"ms_per_step": round(single_ms, 4), "modexps_per_s": round(3 * BATCH * N / (single_ms * 1e-3), 1),
                                         "what": "the same steps on ONE lane, 10 steps behind the timed region (launches do not "
                                                 "overlap in it): roofline.lone_launch and other_kernels are from this pass"}
        # The roofline block describes the dominant kernel AS IT RAN in the timed region.  A lone launch (one lane) is
        # kept beside it.  Under the adaptive policy a decrypt queued beside a busy neighbour lane runs the
        # sequential-halves kernel on HALF the chip (its workgroups claim whole CUs, 128 of 256): its HIP-event duration
        # is the time it holds that half, so its rate is priced against half the peak (chip_share).
        lone = {k: r[k] for k in ("kernel", "kernel_ms", "achieved", "frac", "frac_useful", "executed_mac32_per_launch",
                                  "canonical_achieved", "canonical_frac")}
        lone["what"] = "the paired full-chip kernel a lone caller gets (nothing queued on the other lanes), single-lane pass"
        r["lone_launch"] = lone
        dec_timed = [(f, ms) for k, f, ms in timed_forms if k == K_MODEXP]
        if dec_timed:
            top = max({f for f, _ in dec_timed}, key=lambda f: sum(1 for g, _ in dec_timed if g == f))
            ms_top = float(np.mean([ms for f, ms in dec_timed if f == top]))
            name, per_exp = decrypt_kernel(sk, BATCH, nw, KEY_BITS, busy_for_form(top))
            # CU claim: 128 workgroups on 128 of the 256 CUs -- with TWO lanes a launch holds its half from its first
            # wavefront to its last; with more lanes a launch also waits for a free half inside its event span.  The one-lane
            # product-scanning kernel is 64 workgroups: with FOUR lanes each launch holds a quarter of the chip
            share = 0.5 if (top == 18 and nfl == 2) else (0.25 if (top == 28 and nfl == 4) else None)
            if top & 8:
                r["useful_mac32_per_launch"] = per_exp * 2 * BATCH
                r["useful_note"] = ("the one-lane product-scanning kernel executes only products the split form needs (and takes "
                                    "q*n_0 of every reduction column as a shift): executed = useful")
            r["kernel"] = (f"{name} ({FORM_NAMES.get(top, top)}; CRT-decrypt leg: {2 * BATCH} half-width modexps per launch; "
                           f"{sum(1 for g, _ in dec_timed if g == top)} of the {len(dec_timed)} decrypt launches of the timed region)")
            r["executed_mac32_per_launch"] = per_exp * 2 * BATCH
            r["kernel_ms"] = round(ms_top, 3)
            r["kernel_ms_basis"] = "mean HIP-event duration of those launches INSIDE the timed region (rocprofv3 kernel-trace agrees)"
            if top & 10:
                pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
                pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
                pk_ = "ps_decrypt" if top & 8 else "seq_decrypt"
                if pmc.get(pk_ + "_hbm_bytes_per_launch"):
                    r["traffic"] = pmc[pk_ + "_hbm_bytes_per_launch"]
                    r["traffic_source"] = pmc_source(pmc, pk_ + "_kernel")
                else:
                    r["traffic"] = None
                    r["traffic_source"] = None
            if share:
                r["chip_share"] = share
                # achieved / peak = frac holds for the line's three fields: the launch's rate scaled to the whole chip
                # (it runs on `chip_share` of the CUs for kernel_ms; the other share does the neighbour lane's launch)
                r["achieved"] = sig(per_exp * 2 * BATCH / (ms_top * 1e-3) / share / 1e12, 3)
                r["achieved_on_its_share"] = sig(per_exp * 2 * BATCH / (ms_top * 1e-3) / 1e12, 3)
                r["peak_of_share"] = round(PEAK_TMAC32 * share, 2)
                r["frac"] = sig(per_exp * 2 * BATCH / (ms_top * 1e-3) / 1e12 / (PEAK_TMAC32 * share))
                r["frac_useful"] = sig(r["useful_mac32_per_launch"] / (ms_top * 1e-3) / 1e12 / (PEAK_TMAC32 * share))
                r["frac_basis"] = ("EXECUTED multiply-accumulates per launch / HIP-event duration of the launch / (peak x chip_share): "
                                   "the launch holds %d of the 256 CUs (one workgroup per CU by its LDS claim) while the neighbour "
                                   "lanes' launches use the rest; whole-chip accounting of the same leg: roofline.measured_in_flight"
                                   % int(256 * share))
            elif "decrypt_leg" in measured:
                r["achieved"] = sig(measured["decrypt_leg"]["executed_mac32"] / (measured["decrypt_leg"]["wall_ms"] * 1e-3) / 1e12, 3)
                r["frac"] = measured["decrypt_leg"]["frac_executed"]
                r["frac_useful"] = measured["decrypt_leg"]["frac_useful"]
                r["frac_basis"] = ("launches of several lanes share the SIMDs: fraction from the decrypt-only run on the same lanes "
                                   "(roofline.measured_in_flight), executed multiply-accumulates / wall time / peak")
            r["canonical_achieved"] = None
            for kk in ("canonical_frac",):
                r[kk] = sig(r["canonical_mac32_per_launch"] / (ms_top * 1e-3) / 1e12 / (PEAK_TMAC32 * (share or 1.0))) if share else None
    if "decrypt_leg" in measured:
        r["measured_in_flight"] = measured["decrypt_leg"]
        r["chip_ms_per_launch"] = measured["decrypt_leg"]["ms_per_launch"]
        r["chip_ms_note"] = ("chip time one 8192-ciphertext decrypt costs under the overlap of the timed region (decrypt-only run on "
                             "the same lanes: wall / launches); kernel_ms is the HIP-event span of ONE launch on its share of the chip")
    # executed multiply-accumulates of ALL kernels of a step / ms_per_step / peak: the utilisation of the whole step, by
    # the forms that ran in the timed region
    try:
        enc_timed = [f for k, f, _ in timed_forms if k == K_FB]
        enc_top = max(set(enc_timed), key=enc_timed.count) if enc_timed else 0
        dec_timed2 = [f for k, f, _ in timed_forms if k == K_MODEXP]
        dec_top = max(set(dec_timed2), key=dec_timed2.count) if dec_timed2 else 0
        enc_nm, enc_per_elt, _ = encrypt_kernel(pk, BATCH, nw, KEY_BITS, fb["window"] or 13, 1 if enc_top & 2 else 0)
        dec_per_exp = decrypt_kernel(sk, BATCH, nw, KEY_BITS, busy_for_form(dec_top))[1]
        step_exec = dec_per_exp * 2 * BATCH + enc_per_elt * BATCH
        r["step_executed_mac32"] = step_exec
        r["step_executed_frac"] = sig(step_exec / (elapsed / args.steps / N) / 1e12 / PEAK_TMAC32)
        r["step_executed_note"] = ("executed multiply-accumulates of the decrypt (%s) and encrypt (%s) launches of one step / "
                                   "ms_per_step / peak; the CRT kernel's are not counted" % (FORM_NAMES.get(dec_top, dec_top), enc_nm))
    except Exception as e:                                  # noqa: BLE001
        r["step_executed_frac"] = None
        r["step_executed_note"] = repr(e)[:200]
    if N == 1 and not args.no_extras:
        try:
            dec_forms = [f for k, f, _ in timed_forms if k == K_MODEXP]
            top_f = max(set(dec_forms), key=dec_forms.count) if dec_forms else 0
            kname = "hensel_decrypt_ps_kernel" if top_f & 8 else ("hensel_decrypt_seq_kernel" if top_f & 2 else "hensel_decrypt_kernel")
            t_pmc = time.perf_counter()
            tr = traffic_in_run(kname, nfl, 64 * 200)
            if tr:
                r["traffic"] = tr["bytes"]
                r["traffic_source"] = {"how": "measured in THIS run: the step replayed by tools/pmc_step.py under rocprofv3 --pmc FETCH_SIZE and "
                                              "--pmc WRITE_SIZE (a pass of its own each, counters only); bytes per launch = (2 * FETCH_SIZE + "
                                              "WRITE_SIZE) * 1024, the x2 on FETCH per MI355X_MICROARCH.md (HBM)",
                                       "measured_on_kernel": kname, "launches_averaged": tr["launches_averaged"],
                                       "raw_bytes": tr["raw_bytes"], "seconds": round(time.perf_counter() - t_pmc, 1)}
        except Exception as e:                              # noqa: BLE001
            r["traffic_in_run_error"] = repr(e)[:200]
    if "sustained" in measured:
        result["sustained"] = measured["sustained"]
    if "hardened" in measured:
        result["hardened"] = measured["hardened"]
    result["config"]["secret_table_access"] = "masked" if L.pgpu_get_table_gather_policy() else "indexed"
    result["config"]["secret_table_access_note"] = ("indexed: window-table / fixed-base-table ADDRESSES follow the digits of p-1, q-1 and of the "
                                                    "encryption randomness r (include/pgpu.h, SIDE CHANNELS); masked "
                                                    "(pgpu_set_table_gather_policy(1)): every candidate read, one selected -- the `hardened` block "
                                                    "has that step")
    result["pool"] = per_gpu
    # side measurements must never sink the contract line: a failure is reported inside it
    if N == 1 and not args.no_extras:
        try:
            result.update(extras(pa, L, B, pk, sk, n, p, q, hs, m_host, r_host, per_kind))
        except Exception as e:                              # noqa: BLE001
            result["extras_error"] = repr(e)[:400]
    if N == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(n, p, q, hs, m_host, r_host)
        except Exception as e:                              # noqa: BLE001
            result["cpu_baseline"] = {"error": repr(e)[:400]}
    B.free(*[h for pair in sets for h in pair], *all_c, *all_out)
    del pk, sk
    pa.terminate()
    ctypes.CDLL(None).fflush(None)
    bench_line.emit(result)                             # detail -> bench_detail.json + stderr; compact line last on stdout


FORM_NAMES = {0: "full-width", 1: "paired", 2: "sequential-halves", 18: "sequential-halves+cu-claim", 65: "a/b-wavefronts",
              4: "one-lane", 12: "one-lane product-scanning", 28: "one-lane product-scanning+cu-claim"}


def busy_for_form(form):
    """the busy_lanes argument under which pgpu_decrypt_kernel_form_ex names the kernel a recorded form bit pattern ran"""
    return 3 if form & 8 else (1 if form & 2 else 0)


def collect_trace(L, cap):
    """[(kind, form, ms, lane, start_ms)] of the recorded launches of pool entry 0, in launch order: the library's own
    kernel trace (HIP events on the launch streams; start relative to the first recorded launch)"""
    kinds, forms, lanes = (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
    start, dur = (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
    n = L.pgpu_timing_collect_trace(kinds, forms, lanes, start, dur, cap)
    return [(kinds[i], forms[i], dur[i], lanes[i], start[i]) for i in range(n)]


def collect_forms(L, cap):
    """[(kind, form, ms)] of the recorded launches of pool entry 0, in launch order"""
    kinds, forms, kms = (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_double * cap)()
    n = L.pgpu_timing_collect_ex(kinds, forms, kms, cap)
    return [(kinds[i], forms[i], kms[i]) for i in range(n)]


def forms_to_kinds(rec):
    per = {}
    for k, _, ms in rec:
        per.setdefault(k, []).append(ms)
    return per


def forms_summary(rec):
    """launch counts per (kernel kind, form) of a timed region"""
    names = {K_MODEXP: "decrypt_exponentiation", K_CRT: "crt", K_FB: "fixed_base_encrypt", K_MODMUL: "pair_ops"}
    out = {}
    for k, f, _ in rec:
        key = names.get(k, str(k)) + ":" + FORM_NAMES.get(f, str(f))
        out[key] = out.get(key, 0) + 1
    return out


def decrypt_leg_in_flight(L, B, sk, cts, nfl, nw, launches_per_lane=6):
    """The CRT-decrypt leg MEASURED under the overlap the timed region runs with: decrypt-only steps on the same nfl batch
    lanes (resident ciphertexts of the timed region), wall time / launches = the chip time one launch costs.  Fractions
    by the executed count of the kernels that ran (forms recorded per launch) and by the useful count."""
    from pailliercryptolib_amd import _capi
    outs = [None] * nfl

    def run(k):
        for i in range(k):
            ln = i % nfl
            if outs[ln]:
                L.pgpu_batch_destroy(outs[ln])
            outs[ln] = B.op(L.pgpu_batch_decrypt_crt, sk._h, cts[ln])
    run(2 * nfl)
    _capi.check(L.pgpu_synchronize())
    _capi.check(L.pgpu_set_timing(1))
    k = launches_per_lane * nfl
    t0 = time.perf_counter()
    run(k)
    _capi.check(L.pgpu_synchronize())
    wall = time.perf_counter() - t0
    rec = collect_forms(L, 4 * k + 16)
    _capi.check(L.pgpu_set_timing(0))
    for o in outs:
        if o:
            L.pgpu_batch_destroy(o)
    dec = [(f, ms) for kind, f, ms in rec if kind == K_MODEXP]
    executed = 0
    useful = 0
    for f, _ in dec:
        ex = decrypt_kernel(sk, BATCH, nw, KEY_BITS, busy_for_form(f))[1] * 2 * BATCH
        executed += ex
        useful += ex if f & 8 else decrypt_useful_mac32(nw, KEY_BITS) * 2 * BATCH
    per_launch = wall / max(1, len(dec))
    return {"what": "decrypt-only steps (exponentiation + CRT kernel) on the %d batch lanes of the timed region, %d launches; "
                    "wall time / launches = chip time per 8192-ciphertext decrypt under that overlap" % (nfl, len(dec)),
            "launches": len(dec), "wall_ms": round(wall * 1e3, 3), "ms_per_launch": round(per_launch * 1e3, 4),
            "forms": forms_summary(rec),
            "event_ms_per_launch_mean": round(float(np.mean([ms for _, ms in dec])), 3) if dec else None,
            "event_note": "HIP-event span of one launch: it shares the chip with its neighbours for that long",
            "executed_mac32": executed, "useful_mac32": useful,
            "frac_executed": sig(executed / wall / 1e12 / PEAK_TMAC32),
            "frac_useful": sig(useful / wall / 1e12 / PEAK_TMAC32)}


def fixed_base_info(L, pk, dev=0):
    """window, bytes and build time of the fixed-base table the key holds on a pool entry (it is built OUTSIDE the
    timed region, at the key's first encrypts: the headline's encrypt third leans on it)"""
    w, b, ms = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_double()
    L.pgpu_pubkey_fixed_base_info(pk._h, dev, ctypes.byref(w), ctypes.byref(b), ctypes.byref(ms))
    return {"window": w.value, "table_bytes": b.value, "build_ms": round(ms.value, 2)}


def per_gpu_report(L, N, per_kind0):
    """what a first run on real multi-GPU hardware needs to diagnose itself: how the key images travelled, what the
    replication self-check saw, and the decrypt-kernel time of EVERY GPU (HIP events on each GPU's batch stream)"""
    from pailliercryptolib_amd import _capi
    ver, rep = ctypes.c_uint64(), ctypes.c_uint64()
    L.pgpu_replication_stats(ctypes.byref(ver), ctypes.byref(rep))
    out = {"transport": L.pgpu_pool_transport().decode(), "rccl": L.pgpu_rccl_note().decode(),
           "key_image_copies_verified": ver.value, "key_image_copies_repaired": rep.value, "decrypt_kernel_ms": {}}
    for g in range(N):
        per = per_kind0
        if g > 0:
            _capi.check(L.pgpu_set_device(g))
            per = collect_timing(L, 4096)
        if K_MODEXP in per:
            ms = per[K_MODEXP] if K_FB in per else per[K_MODEXP][1::2]
            out["decrypt_kernel_ms"][str(g)] = round(float(np.mean(ms)), 3)
    if N > 1:
        _capi.check(L.pgpu_set_device(0))
    return out


def decrypt_window_products(e):
    """window products of the CRT-decrypt exponentiation incl. its table build (csrc/policy.cpp: pick_decrypt_window):
    5-bit windows, 6-bit ones for exponents of 1280 bits and more"""
    return (e + 5) // 6 + 62 if e >= 1280 else (e + 4) // 5 + 30


def decrypt_useful_mac32(nw, key_bits):
    """USEFUL multiply-accumulates of one half-width exponentiation of the CRT-decrypt leg: what the split form has to
    compute, whatever kernel computes it -- the count of the sequential-halves form (csrc/hensel_seq.hpp), in which no lane
    sits through another's products: a squaring is a*a with its symmetry (L2 (L2 + G) / 2) + 2ab (L2^2) + two reductions
    (2 L2^2), a general product 3 half-width products + two reductions (5 L2^2).  The paired kernel EXECUTES 4 L2^2 / 6 L2^2
    (half A idles through half B's second product and cannot use the symmetry): executed - useful = slots spent on
    products nobody needs."""
    g, l2 = {1024: (2, 20), 2048: (2, 38), 3072: (4, 56), 4096: (4, 72)}[key_bits]
    e = key_bits // 2
    nmul = decrypt_window_products(e) + (2 * nw + e // 64 - 1) // (e // 64) + 2
    return e * (l2 * (l2 + g) // 2 + 3 * l2 * l2) + nmul * 5 * l2 * l2


def decrypt_kernel(sk, count, nw, key_bits, busy_lanes=0):
    """(name, executed MAC32 per half-width exponentiation) of the kernel a CRT decrypt of `count` ciphertexts runs:
    the split form (csrc/hensel.hpp: a residue modulo p^2 as two half-width numbers, L2 limbs each; a squaring is
    4 L2^2 limb products, a general product 6 L2^2) or the full-width modexp_kernel (executed = the canonical count
    within 1 %: 29-bit limbs cost x1.27, symmetric squaring gives x0.77 back)."""
    from pailliercryptolib_amd import _capi
    split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(_capi.lib().pgpu_decrypt_kernel_form_ex(sk._h, count, busy_lanes, ctypes.byref(split), ctypes.byref(lanes),
                                                        ctypes.byref(limbs)))
    e = key_bits // 2
    if not split.value:
        return f"modexp_kernel<Geo<{lanes.value},{limbs.value}>>", algorithmic_mac32(key_bits, e)
    if split.value == 4:       # a whole exponentiation per lane by product scanning (csrc/hensel_ps.hpp): per squaring
        # a*a with its symmetry, 2ab, and two reductions of L2 (L2 - 1) products (q*n_0 is a shift); a general product
        # 3 L2^2 + 2 L2 (L2 - 1); the entry runs one single and one pair product per chunk, the exit the same
        l2 = limbs.value
        red = l2 * (l2 - 1)
        nmul = decrypt_window_products(e) + 2 + 1
        return (f"hensel_decrypt_ps_kernel<{l2},{29 if l2 == 19 else 28}>", e * (l2 * (l2 + 1) // 2 + l2 * l2 + 2 * red) + nmul * (3 * l2 * l2 + 2 * red)
                + 3 * (l2 * l2 + red))
    seq = split.value == 2
    l2 = (lanes.value if seq else lanes.value // 2) * limbs.value
    if _capi.lib().pgpu_get_secret_exponent_policy():          # sliding schedule of p-1: ~e/7 products, 32 odd powers
        nsq, nmul = e, e // 7 + 32
    else:                                                      # 5-bit fixed window
        nsq, nmul = e, decrypt_window_products(e)
    nmul += (2 * nw + e // 64 - 1) // (e // 64) + 2            # ciphertext chunks in, exit products
    if seq:    # both halves in the same lanes: the a*a of a squaring uses its symmetry, L2 (L2 + lanes) / 2 products
        sq = l2 * (l2 + lanes.value) // 2 + 3 * l2 * l2
        return f"hensel_decrypt_seq_kernel<{lanes.value},{limbs.value}>", nsq * sq + nmul * 5 * l2 * l2
    return f"hensel_decrypt_kernel<{lanes.value // 2},{limbs.value}>", nsq * 4 * l2 * l2 + nmul * 6 * l2 * l2


def nondjn_executed(key_bits, l2=72):
    """executed multiply-accumulates of r^n * (1 + n*m) per element through hensel_modexp_kernel<4,18> (2048-bit keys)"""
    e = key_bits
    nsq, nmul = e - 1, e // 7 + 32 + 4 + 2          # squarings; window products + odd-power table + entry chunks + exit
    return nsq * 4 * l2 * l2 + nmul * 6 * l2 * l2


def fixed_window(exp_bits):
    """the library's window for per-element exponents (csrc/capi.cpp: pick_window): w in 1..5 that minimises the table
    products 2^w - 2 plus the window products ceil(e / w)"""
    return min(range(1, 6), key=lambda w: ((1 << w) - 2) + (exp_bits + w - 1) // w)


def ctpt_block(pk, shard, e_bits, t_mul, total, me_ms, mac_canonical, pmc):
    """config 5 (ii): CT x PT with e_bits-bit plaintexts -- executed and useful multiply-accumulates of the kernel that ran
    (pair rows in and out: no entry or exit products).  Fixed window w: 2^w - 2 table products, then per remaining window
    w squarings and one product."""
    from pailliercryptolib_amd import _capi
    split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(_capi.lib().pgpu_modexp_n2_kernel_form(pk._h, shard, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
    w = fixed_window(e_bits)
    nwin = (e_bits + w - 1) // w
    nsq, nmul = w * (nwin - 1), (1 << w) - 2 + (nwin - 1)
    out = {"ms_per_step": round(t_mul * 1e3, 3), "modexps_per_s": round(total / t_mul, 1),
           "kernel": modexp_n2_kernel(pk, shard, True), "kernel_ms": round(me_ms, 3),
           "schedule": "fixed window w = %d: %d squarings + %d products per element" % (w, nsq, nmul)}
    if split.value:
        g = lanes.value if split.value == 2 else lanes.value // 2
        l2 = g * limbs.value
        useful = nsq * (l2 * (l2 + g) // 2 + 3 * l2 * l2) + nmul * 5 * l2 * l2
        executed = useful if split.value == 2 else nsq * 4 * l2 * l2 + nmul * 6 * l2 * l2
        out.update({"executed_mac32_per_launch": executed * shard, "useful_mac32_per_launch": useful * shard,
                    "frac": sig(executed * shard / (me_ms * 1e-3) / 1e12 / PEAK_TMAC32),
                    "frac_useful": sig(useful * shard / (me_ms * 1e-3) / 1e12 / PEAK_TMAC32),
                    "frac_basis": "EXECUTED multiply-accumulates per launch / kernel time (HIP events) / peak"})
    out["canonical_mac32_per_launch"] = mac_canonical
    out["canonical_frac"] = sig(mac_canonical / (me_ms * 1e-3) / 1e12 / PEAK_TMAC32)
    out["canonical_note"] = ("SURVEY 8(d) count (full-width products, squarings as products, w = 2 for short exponents) / time / peak: "
                             "not a utilisation (the split form executes about half of it); the utilisation figure is frac")
    out["traffic"] = pmc.get("ct_mul_hbm_bytes_per_launch")
    out["traffic_source"] = pmc_source(pmc, "ct_mul_kernel") if pmc.get("ct_mul_hbm_bytes_per_launch") else None
    return out


def modexp_n2_kernel(pk, count, per_element_exponents=False):
    """name of the kernel an exponentiation modulo n^2 with per-element bases runs (CT x PT: per-element exponents on
    resident rows; the non-DJN obfuscator: the host's schedule of n)"""
    from pailliercryptolib_amd import _capi
    split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(_capi.lib().pgpu_modexp_n2_kernel_form(pk._h, count, ctypes.byref(split), ctypes.byref(lanes),
                                                       ctypes.byref(limbs)))
    if split.value == 2:        # both halves of a residue in the same lanes: CT x PT of resident batches only
        if per_element_exponents:
            return f"hensel_modexp_seq_kernel<{lanes.value},{limbs.value}>"
        return f"hensel_modexp_kernel<{lanes.value},{limbs.value}>"
    if split.value:
        return f"hensel_modexp_kernel<{lanes.value // 2},{limbs.value}>"
    return f"modexp_kernel<Geo<{lanes.value},{limbs.value}>>"


def encrypt_kernel(pk, count, nw, key_bits, fbw, busy_lanes=0):
    """(name, executed MAC32 per element, note) of the fixed-base DJN encrypt kernel: full-width products (2 s^2 + s,
    s = 4096/32) or pair products of the split form (6 L2^2 limb products each issued; 5 L2^2 with both halves of a
    residue in the same lanes), plus the exit: onto a pair row two half-width products (4 L2^2), back to full-width words
    one more pair product and two full-width products."""
    from pailliercryptolib_amd import _capi
    L = _capi.lib()
    split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(L.pgpu_encrypt_kernel_form_ex(pk._h, nw, count, busy_lanes, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
    nprod = (key_bits // 2 + fbw - 1) // fbw - 1           # table products
    s = 2 * key_bits // 32
    if not split.value:
        return (f"fb_encrypt_kernel<Geo<{lanes.value},{limbs.value}>>", (2 * s * s + s) * (nprod + 2),
                "fixed-base windowing w=%d: hs^r as %d table products, no squarings" % (fbw, nprod))
    if split.value == 2:
        l2 = lanes.value * limbs.value
        return (f"hensel_fb_encrypt_seq_kernel<{lanes.value},{limbs.value}>", 5 * l2 * l2 * nprod + 4 * l2 * l2,
                "fixed-base windowing w=%d in split form, both halves of a residue in the same lanes: hs^r as %d pair "
                "products (no squarings), 1 + n*m as two half-width products, result stays a pair row" % (fbw, nprod))
    l2 = lanes.value // 2 * limbs.value
    if os.environ.get("PGPU_PAIR_ROWS", "1") != "0":
        return (f"hensel_fb_encrypt_kernel<{lanes.value // 2},{limbs.value}>", 6 * l2 * l2 * nprod + 4 * l2 * l2,
                "fixed-base windowing w=%d in split form: hs^r as %d pair products (no squarings), 1 + n*m as two "
                "half-width products, result stays a pair row" % (fbw, nprod))
    return (f"hensel_fb_encrypt_kernel<{lanes.value // 2},{limbs.value}>",
            6 * l2 * l2 * (nprod + 1) + 2 * l2 * l2 + 2 * (2 * s * s + s),
            "fixed-base windowing w=%d in split form: hs^r as %d pair products (no squarings), exit by 1 + n*m, two "
            "full-width products back to c*R mod n^2" % (fbw, nprod))


def headline(args, world, elapsed, per_kind, nw, pw, parallelism, dec_kernel=None, enc_kernel=None, fb=None):
    """the contract line (metric / value / roofline of the dominant kernel) from the timed region's numbers"""
    fixed_base = K_FB in per_kind
    enc_ms = float(np.mean(per_kind[K_FB])) if fixed_base else None
    modexp_ms = per_kind.get(K_MODEXP, [])
    if not fixed_base:          # generic path: encrypt and decrypt both launch modexp_kernel, alternating
        enc_ms = float(np.mean(modexp_ms[0::2]))
        modexp_ms = modexp_ms[1::2]
    dec_ms = float(np.mean(modexp_ms))
    crt_ms = float(np.mean(per_kind[K_CRT]))
    modexps_per_step = 3 * BATCH * world
    value = modexps_per_step * args.steps / elapsed
    mac_enc = algorithmic_mac32(2 * KEY_BITS, KEY_BITS // 2) * BATCH        # 41.48 M * 8192
    mac_dec = 2 * algorithmic_mac32(KEY_BITS, KEY_BITS // 2) * BATCH        # 20.82 M * 8192
    alg_bytes_dec = (2 * nw * 8 + nw * 8) * BATCH                           # c + m = 768 B/elt
    alg_bytes_enc = (nw * 8 + pw * 8 + 2 * nw * 8) * BATCH                  # m + r + c = 896 B/elt
    canonical = mac_dec / (dec_ms * 1e-3) / 1e12
    dec_name, dec_exec = dec_kernel or (f"modexp_kernel<{geo_name(nw, KEY_BITS, 2 * BATCH)}>",
                                        algorithmic_mac32(KEY_BITS, KEY_BITS // 2))
    # THE roofline fraction is the EXECUTED one: multiply-accumulates the kernel really issues / time / peak.  The
    # canonical (SURVEY 8d square-and-multiply) count is kept beside it: the split form executes 0.77x of it, the
    # fixed-base encrypt 1/15 -- by that count a step exceeds the machine's peak (canonical_step_frac > 1), which says
    # that the step does not DO the canonical work, not that the hardware is being used better than it can be.
    achieved = dec_exec * 2 * BATCH / (dec_ms * 1e-3) / 1e12
    step_ms_gpu = elapsed / args.steps * 1e3
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
    # encrypt leg: with fixed-base tables the kernel EXECUTES far fewer multiplications than the
    # canonical square-and-multiply count, so its honest ALU fraction uses the executed count
    fbw = int(os.environ.get("PGPU_FB_WINDOW", "13"))
    s4096 = 2 * KEY_BITS // 32
    if fixed_base:
        nmul = (KEY_BITS // 2 + fbw - 1) // fbw + 1            # nwin-1 table products + g^m + exit
        mac_enc_exec = (2 * s4096 * s4096 + s4096) * nmul * BATCH
        enc_name = f"fb_encrypt_kernel<{geo_name(2 * nw, 2 * KEY_BITS, BATCH)}>"
        enc_note = "fixed-base windowing w=%d: hs^r as %d table products, no squarings" % (fbw, nmul - 2)
        if enc_kernel:
            enc_name, per_elt, enc_note = enc_kernel
            mac_enc_exec = per_elt * BATCH
    else:
        mac_enc_exec = mac_enc
        enc_name = f"modexp_kernel<{geo_name(2 * nw, 2 * KEY_BITS, BATCH)}> (encrypt)"
        enc_note = "generic square-and-multiply"
    return {
        "metric": "2048-bit modexps/sec (encrypt+decrypt)",
        "value": round(value, 1),
        "unit": "modexps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": "k=2048 ISO/IEC 18033-6 key, DJN: encrypt batch=8192 (mod n^2, 4096 b, e=1024 b) "
                        "+ CRT decrypt batch=8192 (2x mod p^2/q^2, 2048 b, e=1024 b) per GPU per step",
            "batch_per_gpu": BATCH, "modexps_per_step_per_gpu": 3 * BATCH,
            "parallelism": parallelism,
            "elements_per_s": round(BATCH * world * args.steps / elapsed, 1),
        },
        "roofline": {
            "bound": "int-alu",
            "bound_note": "VALU issue rate: v_mad_u64_u32 issues at full rate, one wave64 instruction per 4 cycles per SIMD "
                          "= 39.32 T MAC32/s at 2.4 GHz (38.35 measured, profiles/r01_ubench_mad_peak.txt); "
                          "neither hbm nor mfma binds this path: integer carry-chain work, HBM at 1e-4 of peak",
            "kernel": f"{dec_name} (CRT-decrypt leg: {2 * BATCH} half-width "
                      "modexps per launch; the dominant kernel of the step)",
            "achieved": sig(achieved, 3),
            "peak": PEAK_TMAC32,
            "unit": "TMAC32/s",
            "frac": sig(achieved / PEAK_TMAC32),
            "frac_basis": "EXECUTED multiply-accumulates (limb products of 29-bit limbs counted as MAC32 one for one) "
                          "per launch / kernel time / peak; two significant digits (box-to-box spread 2-4 %)",
            "traffic": pmc.get("modexp_decrypt_hbm_bytes_per_launch"),
            "traffic_source": pmc_source(pmc, "modexp_decrypt_kernel"),
            "kernel_ms": round(dec_ms, 3),
            "executed_mac32_per_launch": dec_exec * 2 * BATCH,
            # the SURVEY 8(d) count, whatever the kernel executes: full-width Montgomery products, squarings counted
            # as products, 5-bit window
            "canonical_mac32_per_launch": mac_dec,
            "canonical_achieved": sig(canonical, 3),
            "canonical_frac": sig(canonical / PEAK_TMAC32),
            "canonical_step_frac": sig((mac_dec + mac_enc) / (step_ms_gpu / world * 1e-3) / 1e12 / PEAK_TMAC32),
            "canonical_step_note": "canonical MAC32 of one step (encrypt + decrypt) / ms_per_step / peak: above 1 because "
                                   "the encrypt third runs as fixed-base table look-ups (1/15 of the canonical "
                                   "multiply-accumulates, table built outside the timed region) and the decrypt "
                                   "leg in split form (0.77x); quote decrypt_only_modexps_per_s for like-for-like",
            "algorithmic_bytes_per_launch": alg_bytes_dec,
            "hbm_achieved_GBs": round(alg_bytes_dec / (dec_ms * 1e-3) / 1e9, 3),
            "hbm_peak_GBs": HBM_PEAK_GBS,
            "hbm_frac": round(alg_bytes_dec / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
            "other_kernels": {
                "crt_kernel<Geo<8,9>>": {"ms": round(crt_ms, 4)},
                enc_name: {
                    "ms": round(enc_ms, 4),
                    "canonical_mac32_per_launch": mac_enc,
                    "executed_mac32_per_launch": mac_enc_exec,
                    "frac": sig(mac_enc_exec / (enc_ms * 1e-3) / 1e12 / PEAK_TMAC32),
                    "canonical_frac": sig(mac_enc / (enc_ms * 1e-3) / 1e12 / PEAK_TMAC32),
                    "fixed_base_table": fb,
                    "algorithmic_bytes_per_launch": alg_bytes_enc,
                    "traffic": pmc.get("fb_encrypt_hbm_bytes_per_launch"),
                    "traffic_source": pmc_source(pmc, "fb_encrypt_kernel"),
                    "note": enc_note,
                },
            },
        },
        # the conservative reading of the headline: the CRT-decrypt leg alone (no fixed-base shortcut in it)
        "decrypt_only_modexps_per_s": round(2 * BATCH / ((dec_ms + crt_ms) * 1e-3), 1),
    }


def traffic_in_run(kernel_substr, lanes, min_grid):
    """HBM bytes per launch of the kernel whose name contains kernel_substr, MEASURED in this run: the headline step replayed
    by tools/pmc_step.py under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, one pass each (counters only, no trace
    domain), bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 averaged over the full-size launches -- the x2 on FETCH per
    MI355X_MICROARCH.md (HBM): gfx950 tallies wide reads at half their size.  None when rocprofv3 is not on PATH, the
    passes fail or take too long (the committed constant of profiles/pmc_summary.json stays in the line then)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe or os.environ.get("BENCH_NO_PMC") == "1":
        return None
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                            os.path.join(ROOT, "tools", "pmc_step.py"), str(lanes), "8"],
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150, check=True)
            got = []
            for fn in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
                for row in csv.DictReader(open(fn)):
                    if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter and int(row["Grid_Size"]) >= min_grid:
                        got.append(float(row["Counter_Value"]))
            if not got:
                return None
            vals[counter] = (sum(got) / len(got), len(got))
        except Exception:                                   # noqa: BLE001
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"bytes": (2 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024,
            "raw_bytes": (vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024, "launches_averaged": vals["FETCH_SIZE"][1]}


def pmc_source(pmc, kernel_key):
    """where a `traffic` figure comes from: PMC counters cannot be read inside bench.py (rocprofv3 owns them), so the
    number is the per-launch average of a committed rocprofv3 --pmc pass (its own run, no trace domains) of THIS command
    on the build named there -- the same on every box until that pass is repeated"""
    if not pmc:
        return None
    return {"file": "profiles/pmc_summary.json", "measured_on_kernel": pmc.get(kernel_key), "how": pmc.get("source"),
            "build": pmc.get("build"), "collected": pmc.get("collected"),
            "note": "constant read from the committed summary (its build and collection date named here), not measured in this run"}


def best_of(fn, reps):
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def extras(pa, L, B, pk, sk, n, p, q, hs, m_host, r_host, per_kind):
    """SURVEY 8(d) side measurements, N = 1, outside the timed region (VERDICT r01 'next' item 3)."""
    from pailliercryptolib_amd import _capi
    nw, pw = m_host.shape[1], r_host.shape[1]
    out = {}
    # (0) the resident step with 1..4 batches in flight, ~0.8 s each (sustained rates; the headline's --in-flight is one of them)
    try:
        out["batches_in_flight_sweep"] = lanes_sweep(L, B, pk, sk, m_host, r_host)
    except Exception as e:                                  # noqa: BLE001
        out["batches_in_flight_sweep"] = {"error": repr(e)[:300]}
    time.sleep(0.15)
    # (1) the same step through the synchronous host-pointer entry points: H2D + kernels + D2H per call
    c_host = np.empty((BATCH, 2 * nw), dtype=np.uint64)
    d_host = np.empty((BATCH, nw), dtype=np.uint64)

    def e2e_enc():
        _capi.check(L.pgpu_paillier_encrypt(pk._h, ptr(m_host), nw, nw, ptr(r_host), pw, pw, 64 * pw, ptr(c_host), BATCH))

    def e2e_dec():
        _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, ptr(c_host), ptr(d_host), BATCH))
    e2e_enc(); e2e_dec()
    te, td = best_of(e2e_enc, 5), best_of(e2e_dec, 5)
    assert np.array_equal(d_host, m_host)
    out["end_to_end"] = {"what": "pgpu_paillier_encrypt + pgpu_paillier_decrypt_crt on caller-owned PAGEABLE host arrays, one "
                                 "synchronous caller (H2D + kernels + D2H inside each call; staging through the library's "
                                 "pinned bounce buffers)",
                         "encrypt_ms": round(te * 1e3, 3), "decrypt_ms": round(td * 1e3, 3),
                         "modexps_per_s": round(3 * BATCH / (te + td), 1),
                         "floor_note": "a synchronous call cannot end before its kernels do: the kernels of one encrypt + one "
                                       "decrypt call are %.2f ms of the %.2f ms measured"
                                       % (float(np.mean(per_kind[K_FB])) + float(np.mean(per_kind[K_MODEXP])) + float(np.mean(per_kind[K_CRT])),
                                          (te + td) * 1e3)}
    # (1b) the same calls on buffers from pgpu_host_alloc: the DMA reads / writes the caller's arrays, no staging copy
    try:
        blocks = []

        def pinned(shape):
            nbytes = int(np.prod(shape)) * 8
            pp = ctypes.c_void_p()
            _capi.check(L.pgpu_host_alloc(nbytes, ctypes.byref(pp)))
            blocks.append(pp)
            return pp, np.frombuffer((ctypes.c_uint8 * nbytes).from_address(pp.value), dtype=np.uint64).reshape(shape)
        pm, am = pinned((BATCH, nw)); pr, ar = pinned((BATCH, pw)); pc, ac = pinned((BATCH, 2 * nw)); pd, ad = pinned((BATCH, nw))
        am[:], ar[:] = m_host, r_host

        def pe():
            _capi.check(L.pgpu_paillier_encrypt(pk._h, pm, nw, nw, pr, pw, pw, 64 * pw, pc, BATCH))

        def pdq():
            _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, pc, pd, BATCH))
        pe(); pdq()
        tpe, tpd = best_of(pe, 5), best_of(pdq, 5)
        assert np.array_equal(ad, m_host) and np.array_equal(ac, c_host)
        out["end_to_end_pinned"] = {"what": "the same two calls on caller buffers from pgpu_host_alloc (pinned: the DMA source / target "
                                            "is the caller's array itself), one synchronous caller",
                                    "encrypt_ms": round(tpe * 1e3, 3), "decrypt_ms": round(tpd * 1e3, 3),
                                    "modexps_per_s": round(3 * BATCH / (tpe + tpd), 1)}
        for b in blocks:
            L.pgpu_host_free(b)
    except Exception as e:                                  # noqa: BLE001
        out["end_to_end_pinned"] = {"error": repr(e)[:300]}
    # (1c) TWO synchronous callers (host threads, pageable arrays of their own): what a service with more than one
    # request in flight sees -- one caller's copies run under the other's kernels, and the callers see each other through
    # the lane activity stamps, so their launches take the half-chip forms of the adaptive policy and run side by side
    # (the reference's own tests call encrypt / decrypt from four OpenMP threads, test_cryptography.cpp:45-57)
    for variant in ("pageable", "pinned"):
        key = "end_to_end_two_callers" + ("" if variant == "pageable" else "_pinned")
        try:
            out[key] = two_callers(L, pk, sk, m_host, r_host, variant)
        except Exception as e:                              # noqa: BLE001
            out[key] = {"error": repr(e)[:300]}
    # ... and FOUR (round 5): each caller sees three others, so its launches take the quarter-chip forms (one-lane decrypt
    # with a CU claim: 14.5 ms per call, four side by side) -- word ciphertexts through the pair-row conversion on the way in
    time.sleep(0.1)
    for variant in ("pageable", "pinned"):
        key = "end_to_end_four_callers" + ("" if variant == "pageable" else "_pinned")
        try:
            out[key] = two_callers(L, pk, sk, m_host, r_host, variant, reps=5, ncall=4)
        except Exception as e:                              # noqa: BLE001
            out[key] = {"error": repr(e)[:300]}
    # (1d) ONE thread pipelining host-to-host steps over two (and four) batch lanes
    for ln in (2, 4):
        try:
            out["end_to_end_pipelined" + ("" if ln == 2 else "_4_lanes")] = pipelined_e2e(L, pk, sk, m_host, r_host, lanes=ln)
        except Exception as e:                              # noqa: BLE001
            out["end_to_end_pipelined" + ("" if ln == 2 else "_4_lanes")] = {"error": repr(e)[:300]}
    time.sleep(0.15)   # (the callers' lanes stay marked active for 50 ms: the lone-caller measurements below start clean)
    # (2) the API-visible timing of the reference's own benchmark: ipcl::PublicKey::encrypt / PrivateKey::decrypt with
    # std::vector<BigNumber> in and out (benchmark/bench_cryptography.cpp:73-121)
    try:
        out["api_level"] = api_level()
    except Exception as e:                                  # noqa: BLE001 -- a side measurement must not sink the line
        out["api_level"] = {"error": str(e)[:300]}
    # (3) config 2 with the plain obfuscator r^n mod n^2 (2048-bit exponent) instead of DJN's hs^r
    pk2 = pa.PublicKey(n, KEY_BITS)
    rng = np.random.default_rng(99)
    r2 = np.frombuffer(rng.bytes(BATCH * nw * 8), dtype=np.uint64).reshape(BATCH, nw).copy()
    r2[:, -1] &= np.uint64((1 << 62) - 1)
    bm, br2 = B.up(m_host), B.up(r2)
    hold = {}

    def nondjn():
        B.free(hold.get("c"))
        hold["c"] = B.op(L.pgpu_batch_encrypt, pk2._h, bm, br2, 64 * nw)
        _capi.check(L.pgpu_synchronize())
    nondjn()
    tn = best_of(nondjn, 3)
    dec = B.op(L.pgpu_batch_decrypt_crt, sk._h, hold["c"])
    assert np.array_equal(B.down(dec), m_host)
    mac = algorithmic_mac32(2 * KEY_BITS, KEY_BITS) * BATCH
    out["config2_nondjn"] = {"what": "encrypt batch=8192 with the non-DJN obfuscator r^n mod n^2 (e = n, 2048 b), resident",
                             "kernel": modexp_n2_kernel(pk2, BATCH),
                             "encrypt_ms": round(tn * 1e3, 3), "encrypts_per_s": round(BATCH / tn, 1),
                             # executed: sliding-window schedule of the PUBLIC exponent n (w = 6: ~e/7 + 32 products), paired split
                             # form with L2 = 72 limbs per half: 4 L2^2 per squaring, 6 L2^2 per product, + entry / exit products
                             "executed_mac32_per_launch": nondjn_executed(KEY_BITS) * BATCH,
                             "frac": sig(nondjn_executed(KEY_BITS) * BATCH / tn / 1e12 / PEAK_TMAC32),
                             "frac_basis": "EXECUTED multiply-accumulates / call time (launch + synchronise) / peak",
                             "canonical_frac": sig(mac / tn / 1e12 / PEAK_TMAC32),
                             "canonical_note": "SURVEY 8(d) count / time / peak: not a utilisation (see roofline.canonical_step_note)",
                             "step_modexps_per_s_with_it": round(
                                 3 * BATCH / (tn + (np.mean(per_kind[K_MODEXP]) + np.mean(per_kind[K_CRT])) * 1e-3), 1)}
    B.free(hold.get("c"), dec, bm, br2)
    # (4) the same step with the opt-in sliding-window schedules of p-1 / q-1 (include/pgpu.h, SIDE CHANNELS): ~5 % fewer
    # multiplications in the decrypt leg, key-dependent operation sequence -- not the default, reported for reference
    from pailliercryptolib_amd.limbs import limbs_to_ints  # noqa: F401
    old = L.pgpu_get_secret_exponent_policy()
    if old == 0:
        _capi.check(L.pgpu_set_secret_exponent_policy(1))
        bm2, br = B.up(m_host), B.up(r_host)
        st = {}

        def step():
            B.free(st.get("c"), st.get("o"))
            st["c"] = B.op(L.pgpu_batch_encrypt, pk._h, bm2, br, 64 * pw)
            st["o"] = B.op(L.pgpu_batch_decrypt_crt, sk._h, st["c"])
        for _ in range(3):                     # (the GPU has idled through the host-side measurements above)
            step()
        _capi.check(L.pgpu_synchronize())
        _capi.check(L.pgpu_set_timing(1))
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        _capi.check(L.pgpu_synchronize())
        dt = (time.perf_counter() - t0) / 10
        per = collect_timing(L, 64)
        _capi.check(L.pgpu_set_timing(0))
        _capi.check(L.pgpu_set_secret_exponent_policy(old))
        assert np.array_equal(B.down(st["o"]), m_host)
        dec_ms = float(np.mean(per[K_MODEXP]))
        mac_dec = 2 * algorithmic_mac32(KEY_BITS, KEY_BITS // 2) * BATCH
        out["sliding_window_policy"] = {"what": "pgpu_set_secret_exponent_policy(PGPU_EXP_SLIDING): host-built sliding-window "
                                                "schedules of p-1 / q-1 (key-dependent operation sequence; opt-in)",
                                        "modexps_per_s": round(3 * BATCH / dt, 1), "ms_per_step": round(dt * 1e3, 4),
                                        "decrypt_kernel_ms": round(dec_ms, 4),
                                        "decrypt_kernel_canonical_frac": sig(mac_dec / (dec_ms * 1e-3) / 1e12 / PEAK_TMAC32)}
        B.free(st.get("c"), st.get("o"), bm2, br)
    # (4b) the opt-in masked table gather (include/pgpu.h, SIDE CHANNELS): every window-table entry is read and the wanted
    # one selected -- what the reference's mbx_exp_mb8 does; cost of the decrypt launch with it
    try:
        _capi.check(L.pgpu_set_table_gather_policy(1))
        bm3, br3 = B.up(m_host), B.up(r_host)
        c3 = B.op(L.pgpu_batch_encrypt, pk._h, bm3, br3, 64 * pw)
        hold3 = {}

        def dec_masked():
            B.free(hold3.get("o"))
            hold3["o"] = B.op(L.pgpu_batch_decrypt_crt, sk._h, c3)
            _capi.check(L.pgpu_synchronize())
        dec_masked()
        tm = best_of(dec_masked, 5)
        assert np.array_equal(B.down(hold3["o"]), m_host)
        _capi.check(L.pgpu_set_table_gather_policy(0))
        dec_masked()
        ti = best_of(dec_masked, 5)
        out["masked_table_gather"] = {"what": "pgpu_set_table_gather_policy(1): CRT decrypt of the batch with every window-table "
                                              "entry read and selected (address stream independent of p-1 / q-1) vs indexed",
                                      "decrypt_ms_masked": round(tm * 1e3, 3), "decrypt_ms_indexed": round(ti * 1e3, 3)}
        # the same policy on the DJN encrypt (round 4): hs^r over the small-window table, every entry of a window read and selected
        hold4 = {}

        def enc_once():
            B.free(hold4.get("c"))
            hold4["c"] = B.op(L.pgpu_batch_encrypt, pk._h, bm3, br3, 64 * pw)
            _capi.check(L.pgpu_synchronize())
        enc_once()
        tei = best_of(enc_once, 5)
        ref_c = B.down(hold4["c"])
        _capi.check(L.pgpu_set_table_gather_policy(1))
        enc_once(); enc_once()
        tem = best_of(enc_once, 5)
        same = bool(np.array_equal(B.down(hold4["c"]), ref_c))
        _capi.check(L.pgpu_set_table_gather_policy(0))
        B.free(hold4.get("c"))
        if not same:
            raise RuntimeError("masked fixed-base encrypt differs from the indexed one")
        out["masked_fixed_base_encrypt"] = {"what": "pgpu_set_table_gather_policy(1) on the DJN encrypt of the batch: hs^r as 255 products "
                                                    "over a 4-bit-window table, all 16 entries of a window read and the wanted one selected "
                                                    "(address stream independent of r), vs 78 products (w = 13) addressed by the digits of r; "
                                                    "call time incl. launch and synchronise, results identical",
                                            "encrypt_ms_masked": round(tem * 1e3, 3), "encrypt_ms_indexed": round(tei * 1e3, 3)}
        B.free(hold3.get("o"), c3, bm3, br3)
    except Exception as e:                                  # noqa: BLE001
        _capi.check(L.pgpu_set_table_gather_policy(0))
        out.setdefault("masked_table_gather", {"error": str(e)[:300]})
        out.setdefault("masked_fixed_base_encrypt", {"error": str(e)[:300]})
    # (5) two batches in flight: consecutive steps alternate between two HIP streams (`_dev` entry points), so that the
    # encrypt launch of step i+1 and the decrypt launches of steps i / i+1 share the SIMDs -- how a pool entry with its
    # two worker lanes actually runs under load.  A lone wavefront on a SIMD issues every ~4.6 cycles, two every ~4.3
    # (DESIGN.md section 2): the same work, 0-8 % sooner.  (The headline runs two resident batches in flight as well, on the
    # library's own batch lanes; this is the same thing through caller-owned streams and word ciphertexts.)
    try:
        out["two_streams"] = two_streams(L, pk, sk, m_host, r_host)
    except Exception as e:                                  # noqa: BLE001
        out["two_streams"] = {"error": str(e)[:300]}
    # (6) opt-in PGPU_SEQ_DECRYPT=3, the mode for two batch lanes that are both kept busy: each lane's decrypt in the
    # sequential-halves form (csrc/hensel_seq.hpp; 512 wavefronts for 8192 ciphertexts = half the chip, 11 % fewer
    # instructions), workgroups claiming more than half a CU's LDS so that the two launches spread over all CUs
    try:
        out["sequential_halves_two_lanes"] = seq_two_lanes(L, B, pk, sk, m_host, r_host)
    except Exception as e:                                  # noqa: BLE001
        L.pgpu_debug_set_seq_decrypt(1)
        out["sequential_halves_two_lanes"] = {"error": str(e)[:300]}
    return out


def two_callers(L, pk, sk, m_host, r_host, variant, reps=6, ncall=2):
    import threading
    from pailliercryptolib_amd import _capi
    nw, pw = m_host.shape[1], r_host.shape[1]
    held = []

    def pin(shape):
        nbytes = int(np.prod(shape)) * 8
        pp = ctypes.c_void_p()
        _capi.check(L.pgpu_host_alloc(nbytes, ctypes.byref(pp)))
        held.append(pp)
        return np.frombuffer((ctypes.c_uint8 * nbytes).from_address(pp.value), dtype=np.uint64).reshape(shape)
    bufs = []
    for _ in range(ncall):
        if variant == "pageable":
            bufs.append((m_host.copy(), r_host.copy(), np.empty((BATCH, 2 * nw), dtype=np.uint64), np.empty((BATCH, nw), dtype=np.uint64)))
        else:
            a, b2 = pin((BATCH, nw)), pin((BATCH, pw))
            a[:], b2[:] = m_host, r_host
            bufs.append((a, b2, pin((BATCH, 2 * nw)), pin((BATCH, nw))))
    bar = threading.Barrier(ncall + 1)
    errs = []
    dbg = [[] for _ in range(ncall)]

    def caller(k):
        mm, rr, cc, dd = bufs[k]
        try:
            # (two untimed rounds side by side first: both worker lanes of the device get their staging buffers, stream
            # workspaces and allocator blocks -- one-time costs of tens of ms that a six-round mean would carry)
            bar.wait()
            for it in range(reps + 2):
                if it == 2:
                    bar.wait()
                ta = time.perf_counter()
                _capi.check(L.pgpu_paillier_encrypt(pk._h, ptr(mm), nw, nw, ptr(rr), pw, pw, 64 * pw, ptr(cc), BATCH))
                tb = time.perf_counter()
                _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, ptr(cc), ptr(dd), BATCH))
                dbg[k].append((round((tb - ta) * 1e3, 2), round((time.perf_counter() - tb) * 1e3, 2)))
        except Exception as e:                              # noqa: BLE001
            errs.append(repr(e))
            try:
                bar.abort()
            except Exception:                               # noqa: BLE001
                pass
    th = [threading.Thread(target=caller, args=(k,)) for k in range(ncall)]
    # (the interpreter's cyclic collector holds the GIL for 40-60 ms once the bench process carries a few million objects,
    # and a caller returning from the library waits for it: both callers showed one 45-58 ms "encrypt" in the same round --
    # none of it inside the library call.  Collected before, switched off for the 0.1 s of the measurement.)
    import gc
    gc.collect()
    gc.disable()
    try:
        for t in th:
            t.start()
        bar.wait()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
    finally:
        gc.enable()
    if os.environ.get("BENCH_DEBUG_TWO"):
        print("two_callers", variant, dbg, file=sys.stderr)
    ok = not errs and all(bool(np.array_equal(b[3], m_host)) for b in bufs)
    for pp in held:
        L.pgpu_host_free(pp)
    if not ok:
        raise RuntimeError("two callers: " + "; ".join(errs)[:200])
    return {"what": "%d host threads, each calling pgpu_paillier_encrypt + pgpu_paillier_decrypt_crt synchronously on %s arrays "
                    "of its own, %d rounds each; aggregate rate" % (ncall, variant, reps),
            "wall_ms": round(wall * 1e3, 3), "ms_per_encrypt_plus_decrypt": round(wall / (reps * ncall) * 1e3, 3),
            "modexps_per_s": round(3 * BATCH * reps * ncall / wall, 1)}


def pipelined_e2e(L, pk, sk, m_host, r_host, lanes=2, steps=24):
    """ONE host thread pipelining whole host-to-host steps over `lanes` batch lanes: per step the plaintexts and the
    randomness are uploaded from the caller's (pinned) arrays, encrypted, decrypted and the result is downloaded into the
    caller's array -- nothing resident between steps.  The thread never waits inside a step: uploads from pinned blocks
    only queue their copy, the operations are asynchronous, pgpu_batch_download_async returns a ticket; a lane's ticket is
    waited for when the lane comes round again.  (The reference's accelerator path has the same shape: HE-QAT submits
    batches and polls for completions, ipcl/mod_exp.cpp:162-440.)"""
    from pailliercryptolib_amd import _capi
    nw, pw = m_host.shape[1], r_host.shape[1]
    held = []

    def pin(shape):
        nbytes = int(np.prod(shape)) * 8
        pp = ctypes.c_void_p()
        _capi.check(L.pgpu_host_alloc(nbytes, ctypes.byref(pp)))
        held.append(pp)
        return np.frombuffer((ctypes.c_uint8 * nbytes).from_address(pp.value), dtype=np.uint64).reshape(shape)
    slots = []
    for _ in range(lanes):
        a, b2, o = pin((BATCH, nw)), pin((BATCH, pw)), pin((BATCH, nw))
        a[:], b2[:] = m_host, r_host
        o[:] = 0
        slots.append({"m": a, "r": b2, "out": o, "ticket": None, "live": []})

    def retire(sl):
        if sl["ticket"] is not None:
            _capi.check(L.pgpu_ticket_wait(sl["ticket"]))
            sl["ticket"] = None
        for h in sl["live"]:
            L.pgpu_batch_destroy(h)
        sl["live"] = []

    def step(i):
        sl = slots[i % lanes]
        retire(sl)                       # the lane's previous step has delivered: its buffers and batches are free
        _capi.check(L.pgpu_set_batch_lane(i % lanes))
        hs_ = []
        for arr, words in ((sl["m"], nw), (sl["r"], pw)):
            h = ctypes.c_void_p()
            _capi.check(L.pgpu_batch_upload(ptr(arr), BATCH, words, words, ctypes.byref(h)))
            hs_.append(h)
        c = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_encrypt(pk._h, hs_[0], hs_[1], 64 * pw, ctypes.byref(c)))
        o = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_decrypt_crt(sk._h, c, ctypes.byref(o)))
        t = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_download_async(o, ptr(sl["out"]), ctypes.byref(t)))
        sl["ticket"] = t
        sl["live"] = hs_ + [c, o]
    try:
        for i in range(2 * lanes):
            step(i)
        for sl in slots:
            retire(sl)
        for sl in slots:
            sl["out"][:] = 0
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        t_issue = time.perf_counter() - t0
        for sl in slots:
            retire(sl)
        wall = time.perf_counter() - t0
        ok = all(bool(np.array_equal(sl["out"], m_host)) for sl in slots)
    finally:
        _capi.check(L.pgpu_set_batch_lane(0))
        for sl in slots:
            try:
                retire(sl)
            except Exception:                               # noqa: BLE001
                pass
        for pp in held:
            L.pgpu_host_free(pp)
    if not ok:
        raise RuntimeError("pipelined end-to-end: results differ")
    return {"what": "ONE host thread, %d batch lanes: per step upload of m and r from pinned caller arrays, encrypt, CRT decrypt, "
                    "asynchronous download into the caller's array (pgpu_batch_download_async); %d steps, nothing resident "
                    "between steps" % (lanes, steps),
            "ms_per_step": round(wall / steps * 1e3, 4), "modexps_per_s": round(3 * BATCH * steps / wall, 1),
            "host_time_in_calls_ms_per_step": round(t_issue / steps * 1e3, 4)}


def lanes_sweep(L, B, pk, sk, m_host, r_host, seconds=0.8):
    from pailliercryptolib_amd import _capi
    pw = r_host.shape[1]
    res = {"what": "encrypt + decrypt steps of 8192 rotating over k batch lanes, ~%.1f s back to back per k, default policies; "
                   "results checked" % seconds}
    for nl in range(1, L.pgpu_batch_lanes() + 1):
        time.sleep(0.12)
        sets = []
        for ln in range(nl):
            _capi.check(L.pgpu_set_batch_lane(ln))
            sets.append((B.up(m_host), B.up(r_host)))
        _capi.check(L.pgpu_set_batch_lane(0))
        st = {"c": [None] * nl, "o": [None] * nl, "i": 0}

        def step():
            k = st["i"] % nl
            st["i"] += 1
            B.free(st["c"][k], st["o"][k])
            st["c"][k] = B.op(L.pgpu_batch_encrypt, pk._h, sets[k][0], sets[k][1], 64 * pw)
            st["o"][k] = B.op(L.pgpu_batch_decrypt_crt, sk._h, st["c"][k])
        for _ in range(2 * nl):
            step()
        _capi.check(L.pgpu_synchronize())
        k = max(nl, int(seconds / 5.2e-3) // nl * nl)
        _capi.check(L.pgpu_set_timing(1))
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        _capi.check(L.pgpu_synchronize())
        dt = time.perf_counter() - t0
        rec = collect_forms(L, 8 * k + 64)
        _capi.check(L.pgpu_set_timing(0))
        ok = all(bool(np.array_equal(B.down(o), m_host)) for o in st["o"])
        B.free(*st["c"], *st["o"], *[h for pair in sets for h in pair])
        if not ok:
            raise RuntimeError("round trip failed with %d lanes" % nl)
        res[str(nl)] = {"steps": k, "ms_per_step": round(dt / k * 1e3, 4), "modexps_per_s": round(3 * BATCH * k / dt, 1),
                        "forms": forms_summary(rec)}
    return res


def seq_two_lanes(L, B, pk, sk, m_host, r_host, steps=20):
    from pailliercryptolib_amd import _capi
    pw = r_host.shape[1]
    sets = []
    for ln in range(2):
        _capi.check(L.pgpu_set_batch_lane(ln))
        sets.append((B.up(m_host), B.up(r_host)))
    _capi.check(L.pgpu_set_batch_lane(0))
    st = {"c": [None, None], "o": [None, None], "i": 0}

    def step():
        k = st["i"] % 2
        st["i"] += 1
        B.free(st["c"][k], st["o"][k])
        st["c"][k] = B.op(L.pgpu_batch_encrypt, pk._h, sets[k][0], sets[k][1], 64 * pw)
        st["o"][k] = B.op(L.pgpu_batch_decrypt_crt, sk._h, st["c"][k])

    def run():
        for _ in range(4):
            step()
        _capi.check(L.pgpu_synchronize())
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        _capi.check(L.pgpu_synchronize())
        return (time.perf_counter() - t0) / steps
    t_def = run()
    L.pgpu_debug_set_seq_decrypt(3)
    try:
        t_seq = run()
        ok = all(bool(np.array_equal(B.down(o), m_host)) for o in st["o"])
    finally:
        L.pgpu_debug_set_seq_decrypt(1)
    B.free(*st["c"], *st["o"], *[h for pair in sets for h in pair])
    if not ok:
        raise RuntimeError("round trip failed")
    return {"what": "two batches in flight, %d steps each: default policy (paired decrypt kernel, 1024 wavefronts per launch) vs "
                    "PGPU_SEQ_DECRYPT=3 (sequential-halves decrypt kernel, 512 wavefronts per launch on half the CUs, "
                    "the two lanes' launches side by side); results checked" % steps,
            "default_ms_per_step": round(t_def * 1e3, 3), "seq3_ms_per_step": round(t_seq * 1e3, 3),
            "default_modexps_per_s": round(3 * BATCH / t_def, 1), "seq3_modexps_per_s": round(3 * BATCH / t_seq, 1)}


def two_streams(L, pk, sk, m_host, r_host, steps=20):
    import torch
    from pailliercryptolib_amd import _capi
    nw, pw = m_host.shape[1], r_host.shape[1]
    d_m = torch.from_numpy(m_host.view(np.int64)).cuda()
    d_r = torch.from_numpy(r_host.view(np.int64)).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    d_c = [torch.empty((BATCH, 2 * nw), dtype=torch.int64, device="cuda") for _ in streams]
    d_o = [torch.empty((BATCH, nw), dtype=torch.int64, device="cuda") for _ in streams]
    torch.cuda.synchronize()

    def run(nstreams):
        def step(i):
            k = i % nstreams
            sp = ctypes.c_void_p(streams[k].cuda_stream)
            _capi.check(L.pgpu_paillier_encrypt_dev(pk._h, d_m.data_ptr(), nw, nw, d_r.data_ptr(), pw, pw, 64 * pw,
                                                    d_c[k].data_ptr(), BATCH, sp))
            _capi.check(L.pgpu_paillier_decrypt_crt_dev(sk._h, d_c[k].data_ptr(), d_o[k].data_ptr(), BATCH, sp))
        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    # the decrypt kernel's full-budget build (291 registers) leaves no room for a second wavefront on its SIMD; its
    # 256-register build does (the library default since round 3): two decrypt launches (or decrypt + encrypt) then share
    # the SIMDs.  The A/B switches the build explicitly and restores the default.
    try:
        L.pgpu_debug_set_packed_decrypt(0)
        t1 = run(1)
        t2 = run(2)
        L.pgpu_debug_set_packed_decrypt(1)
        t1p = run(1)
        t2p = run(2)
    finally:
        L.pgpu_debug_set_packed_decrypt(1)
    ok = all(bool(torch.equal(o, d_m)) for o in d_o)
    if not ok:
        raise RuntimeError("two-stream round trip failed")
    return {"what": "the same step through the `_dev` entry points: one HIP stream vs consecutive steps alternating "
                    "between two streams (two batches in flight on the GPU); 20 steps each, results checked",
            "full_budget_decrypt_build": {"what": "the 291-register build of the decrypt kernel (one wavefront per SIMD; "
                                                  "pgpu_debug_set_packed_decrypt(0), not the default)",
                                          "one_stream_ms_per_step": round(t1 * 1e3, 3), "two_streams_ms_per_step": round(t2 * 1e3, 3),
                                          "two_streams_modexps_per_s": round(3 * BATCH / t2, 1)},
            "one_stream_ms_per_step": round(t1p * 1e3, 3), "two_streams_ms_per_step": round(t2p * 1e3, 3),
            "one_stream_modexps_per_s": round(3 * BATCH / t1p, 1), "two_streams_modexps_per_s": round(3 * BATCH / t2p, 1),
            "packed_decrypt_build": {"what": "the 256-register build of the decrypt kernel (two wavefronts fit a SIMD; the default)",
                                     "one_stream_ms_per_step": round(t1p * 1e3, 3),
                                     "two_streams_ms_per_step": round(t2p * 1e3, 3),
                                     "two_streams_modexps_per_s": round(3 * BATCH / t2p, 1)}}


def api_level():
    from pailliercryptolib_amd import build
    exe = build.build_api_bench()
    r = subprocess.run([exe, "--json", str(BATCH)], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-300:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    d["modexps_per_s"] = round(3 * BATCH / ((d["encrypt_us"] + d["decrypt_us"]) * 1e-6), 1)
    # the same calls from several host threads at once (the reference's tests call the API from an OpenMP team,
    # test_cryptography.cpp:45-57): each thread is synchronous, their calls overlap on the GPU
    d["threads"] = {}
    for t in (2, 4):
        try:
            r = subprocess.run([exe, "--threads", str(t), str(BATCH), "8"], capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            j = json.loads(line)
            d["threads"][str(t)] = {"us_per_encrypt_plus_decrypt": j["us_per_encrypt_plus_decrypt"], "modexps_per_s": j["modexps_per_s"],
                                    "round_trip_ok": j["round_trip_ok"]}
        except Exception as e:                              # noqa: BLE001
            d["threads"][str(t)] = {"error": repr(e)[:200]}
    return d


# ----------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3] / configs[4]: fixed total, sharded over the pool
# ----------------------------------------------------------------------------------------------------------------
def run_config45(args, pa, L, B, N):
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import limbs_to_ints
    sync = lambda: _capi.check(L.pgpu_synchronize())
    if args.config == 4:
        case = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "seeded_vectors.json")))["cases"]
                if c["bits"] == 3072 and c["djn"]][0]
        p, q, hs = int(case["p"], 16), int(case["q"], 16), int(case["hs"], 16)
        n, bits, total = p * q, 3072, 65536
        nw, pw = bits // 64, bits // 128
        pk, sk = pa.PublicKey(n, bits, hs=hs), pa.PrivateKey(p, q)
        m_host, r_host = synth(0, total, nw, pw)
        bm, br = B.up(m_host), B.up(r_host)
        st = {}

        def step():
            B.free(st.get("c"), st.get("o"))
            st["c"] = B.op(L.pgpu_batch_encrypt, pk._h, bm, br, 64 * pw)
            st["o"] = B.op(L.pgpu_batch_decrypt_crt, sk._h, st["c"])
        for _ in range(max(1, args.warmup)):
            step()
        sync()
        _capi.check(L.pgpu_set_timing(1))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        t_issue = time.perf_counter() - t0
        sync()
        elapsed = time.perf_counter() - t0
        per = collect_timing(L, 4 * args.steps + 8)
        _capi.check(L.pgpu_set_timing(0))
        assert np.array_equal(B.down(st["o"]), m_host), "config 4: round trip failed"
        opk = orc.PublicKey(n, bits)
        opk.set_djn(hs)
        assert limbs_to_ints(B.down(st["c"])[:2]) == opk.encrypt(limbs_to_ints(m_host[:2]), limbs_to_ints(r_host[:2]))
        # host-buffer (PCIe-inclusive) variant of the same step
        c_host = np.empty((total, 2 * nw), dtype=np.uint64)
        d_host = np.empty((total, nw), dtype=np.uint64)

        def e2e():
            _capi.check(L.pgpu_paillier_encrypt(pk._h, ptr(m_host), nw, nw, ptr(r_host), pw, pw, 64 * pw, ptr(c_host), total))
            _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, ptr(c_host), ptr(d_host), total))
        e2e()
        t_e2e = best_of(e2e, 2)
        assert np.array_equal(d_host, m_host)
        shard = total // N
        dec_ms = float(np.mean(per[K_MODEXP]))
        mac_dec = 2 * algorithmic_mac32(bits, bits // 2) * shard
        dec_name, dec_exec = decrypt_kernel(sk, shard, nw, bits)
        exec_dec = dec_exec * 2 * shard
        useful_dec = decrypt_useful_mac32(nw, bits) * 2 * shard
        fb4 = fixed_base_info(L, pk)
        enc_name, enc_per_elt, enc_note = encrypt_kernel(pk, shard, nw, bits, fb4["window"] or 13)
        enc_ms = float(np.mean(per[K_FB]))
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
        return {
            "metric": "3072-bit modexps/sec (encrypt+decrypt), batch=65536 sharded over the GPUs",
            "value": round(3 * total * args.steps / elapsed, 1), "unit": "modexps/s", "n_gpus": N, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: k=3072 DJN key, batch=65536 encrypt + CRT decrypt, sharded "
                                   f"contiguously over {N} GPU(s) ({shard} elements each), resident",
                       "batch_per_gpu": shard, "batches_in_flight_per_gpu": 1,
                       "secret_table_access": "masked" if L.pgpu_get_table_gather_policy() else "indexed",
                       "elements_per_s": round(total * args.steps / elapsed, 1),
                       "parallelism": f"in-process device pool x{N} (key images: {L.pgpu_pool_transport().decode()})"},
            "roofline": {"bound": "int-alu", "kernel": f"{dec_name} (CRT-decrypt leg, "
                                                       f"{2 * shard} half-width modexps per launch per GPU)",
                         "achieved": sig(exec_dec / (dec_ms * 1e-3) / 1e12, 3), "peak": PEAK_TMAC32, "unit": "TMAC32/s",
                         "frac": sig(exec_dec / (dec_ms * 1e-3) / 1e12 / PEAK_TMAC32),
                         "frac_basis": "EXECUTED multiply-accumulates per launch / kernel time (HIP events, launches do not "
                                       "overlap in this run) / peak",
                         "executed_mac32_per_launch": exec_dec, "useful_mac32_per_launch": useful_dec,
                         "frac_useful": sig(useful_dec / (dec_ms * 1e-3) / 1e12 / PEAK_TMAC32),
                         "traffic": pmc.get("config4_decrypt_hbm_bytes_per_launch"),
                         "traffic_source": pmc_source(pmc, "config4_decrypt_kernel") if pmc.get("config4_decrypt_hbm_bytes_per_launch") else None,
                         "algorithmic_bytes_per_launch": (4 * L.pgpu_batch_row_limbs(st["c"]) or 2 * nw * 8) * shard + nw * 8 * shard,
                         "canonical_mac32_per_launch": mac_dec,
                         "canonical_frac": sig(mac_dec / (dec_ms * 1e-3) / 1e12 / PEAK_TMAC32),
                         "canonical_note": "SURVEY 8(d) count (full-width products, squarings as products) / time / peak; the "
                                           "utilisation figure is frac",
                         "kernel_ms": round(dec_ms, 4),
                         "other_kernels": {enc_name: {"ms": round(enc_ms, 4), "executed_mac32_per_launch": enc_per_elt * shard,
                                                      "frac": sig(enc_per_elt * shard / (enc_ms * 1e-3) / 1e12 / PEAK_TMAC32),
                                                      "fixed_base_table": fb4, "note": enc_note,
                                                      "traffic": pmc.get("config4_encrypt_hbm_bytes_per_launch"),
                                                      "traffic_source": pmc_source(pmc, "config4_encrypt_kernel") if pmc.get("config4_encrypt_hbm_bytes_per_launch") else None},
                                           "crt_kernel": {"ms": round(float(np.mean(per[K_CRT])), 4)}}},
            "end_to_end": {"what": "the same step from caller-owned host arrays (H2D + kernels + D2H)",
                           "ms_per_step": round(t_e2e * 1e3, 3), "modexps_per_s": round(3 * total / t_e2e, 1)},
            "host_issue_ms_per_step": round(t_issue / args.steps * 1e3, 4),
            "host_issue_note": "time the calling thread spends queueing a step's launches over the %d pool entries; %.2f %% of a step"
                               % (N, 100 * t_issue / elapsed),
        }
    # ---- config 5: 2048-bit CT+CT and CT x PT on 1 M elements ----
    p, q, hs = iso_key()
    n = p * q
    nsq = n * n
    nw = KEY_BITS // 64
    W = 2 * nw
    total = 1 << 20
    pk = pa.PublicKey(n, KEY_BITS, hs=hs)
    rng = np.random.default_rng(5)
    a_host = np.frombuffer(rng.bytes(total * W * 8), dtype=np.uint64).reshape(total, W).copy()
    b_host = np.frombuffer(rng.bytes(total * W * 8), dtype=np.uint64).reshape(total, W).copy()
    a_host[:, -1] &= np.uint64((1 << 60) - 1)        # < n^2 (top word of n^2 is > 2^61 for this key)
    b_host[:, -1] &= np.uint64((1 << 60) - 1)
    e_host = np.frombuffer(rng.bytes(total * 8), dtype=np.uint64).reshape(total, 1).copy() & np.uint64(0xFFFFFFFF)
    ba_plain, bb_plain, be = B.up(a_host), B.up(b_host), B.up(e_host)
    # a resident chain keeps its ciphertexts in the Montgomery domain: bring the operands there once, untimed
    one = B.up(np.array([[1] + [0] * (W - 1)], dtype=np.uint64))
    ba = B.op(L.pgpu_batch_ct_add, pk._h, ba_plain, one)
    bb = B.op(L.pgpu_batch_ct_add, pk._h, bb_plain, one)
    sync()
    st = {}

    def add():
        B.free(st.get("s"))
        st["s"] = B.op(L.pgpu_batch_ct_add, pk._h, ba, bb)

    def mul():
        B.free(st.get("t"))
        st["t"] = B.op(L.pgpu_batch_ct_mul, pk._h, ba, be, 32)
    add(); mul(); sync()
    _capi.check(L.pgpu_set_timing(1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        add()
    t_issue_add = (time.perf_counter() - t0) / args.steps
    sync()
    t_add = (time.perf_counter() - t0) / args.steps
    per_add = collect_timing(L, args.steps + 8)
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps // 4)):
        mul()
    sync()
    t_mul = (time.perf_counter() - t0) / max(1, args.steps // 4)
    per_mul = collect_timing(L, args.steps + 8)
    _capi.check(L.pgpu_set_timing(0))
    idx = [0, 1, total // 2, total - 1]
    av, bv, ev = limbs_to_ints(a_host[idx]), limbs_to_ints(b_host[idx]), [int(x) for x in e_host[idx, 0]]
    assert limbs_to_ints(B.down(st["s"])[idx]) == [x * y % nsq for x, y in zip(av, bv)], "config 5: CT+CT differs"
    assert limbs_to_ints(B.down(st["t"])[idx]) == [pow(x, y, nsq) for x, y in zip(av, ev)], "config 5: CTxPT differs"
    # host-buffer (PCIe-bound) variant of CT+CT
    o_host = np.empty_like(a_host)
    mod = np.array([(nsq >> (64 * i)) & ((1 << 64) - 1) for i in range(W)], dtype=np.uint64)

    def e2e_add():
        _capi.check(L.pgpu_modmul(ptr(a_host), ptr(b_host), W, ptr(mod), W, ptr(o_host), total))
    e2e_add()
    t_e2e = best_of(e2e_add, 2)
    shard = total // N
    mm_ms = float(np.mean(per_add[K_MODMUL]))
    # SURVEY 8(d) / BASELINE.md: "CT add = 65 792 MAC32" = TWO canonical Montgomery products mod n^2 per element (into the
    # domain and the product).  A resident chain executes ONE product: on pair rows (round 3) one PAIR product, three
    # half-width products and two half-width reductions = 6 L2^2 limb products; on Montgomery-form words (round 2,
    # PGPU_PAIR_ROWS=0) one full-width product of 144 limbs, 2 * 144^2.
    mac_add = 2 * (2 * 128 * 128 + 128) * shard
    row_limbs = L.pgpu_batch_row_limbs(st["s"])
    if row_limbs:
        l2 = row_limbs // 2
        split, lanes, limbs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _capi.check(L.pgpu_ct_add_kernel_form(pk._h, shard, ctypes.byref(split), ctypes.byref(lanes), ctypes.byref(limbs)))
        if split.value == 2:   # both halves in the same lanes: a*c, a*d + b*c and two reductions, nothing idle: 5 L2^2
            add_kernel = f"pair_mul_seq_kernel<{lanes.value},{limbs.value}> (CT+CT as one pair product on pair rows, both halves of a residue in the same lanes)"
            exec_add = 5 * l2 * l2 * shard
        else:                  # half A sits through half B's second product: 2 x 3 L2^2 issued
            add_kernel = f"pair_ops_kernel<{lanes.value // 2},{limbs.value}> (PO_MUL: CT+CT as one pair product on pair rows)"
            exec_add = 6 * l2 * l2 * shard
        row_bytes = 4 * row_limbs
    else:
        add_kernel = f"modmul_kernel<{geo_name(W, 2 * KEY_BITS, shard)}> (CT+CT on Montgomery-form words)"
        exec_add = 2 * 144 * 144 * shard
        row_bytes = W * 8
    mac_mul = algorithmic_mac32(2 * KEY_BITS, 32) * shard
    me_ms = float(np.mean(per_mul[K_MODEXP]))
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
    return {
        "metric": "2048-bit CipherText add (modmul mod n^2) elements/sec, batch=1M sharded over the GPUs",
        "value": round(total / t_add, 1), "unit": "modmuls/s", "n_gpus": N, "steps": args.steps, "warmup": 1,
        "ms_per_step": round(t_add * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: k=2048, batch=1M: (i) CT+CT on resident ciphertexts "
                               "(one product each), (ii) CT x PT with 32-bit plaintexts; sharded contiguously over "
                               f"{N} GPU(s) ({shard} elements each)",
                   "batch_per_gpu": shard, "batches_in_flight_per_gpu": 1,
                   "secret_table_access": "none (no secret operand in CT+CT / CT x PT)",
                   "resident_ciphertext_form": ("pair rows (%d limbs)" % row_limbs) if row_limbs else "Montgomery-form words",
                   "parallelism": f"in-process device pool x{N} (key images: {L.pgpu_pool_transport().decode()})"},
        "roofline": {"bound": "int-alu", "kernel": f"{add_kernel}, {shard} per GPU",
                     "achieved": sig(exec_add / (mm_ms * 1e-3) / 1e12, 3), "peak": PEAK_TMAC32, "unit": "TMAC32/s",
                     "frac": sig(exec_add / (mm_ms * 1e-3) / 1e12 / PEAK_TMAC32),
                     "frac_basis": "executed multiply-accumulates per launch / kernel time / peak",
                     "canonical_frac": sig(mac_add / (mm_ms * 1e-3) / 1e12 / PEAK_TMAC32),
                     "traffic": pmc.get("ct_add_pair_mul_hbm_bytes_per_launch") if row_limbs else None,
                     "traffic_source": pmc_source(pmc, "ct_add_kernel") if row_limbs else None,
                     "kernel_ms": round(mm_ms, 4), "executed_mac32_per_launch": exec_add,
                     "algorithmic_bytes_per_launch": 3 * row_bytes * shard,
                     "hbm_achieved_GBs": round(3 * row_bytes * shard / (mm_ms * 1e-3) / 1e9, 1), "hbm_peak_GBs": HBM_PEAK_GBS,
                     "hbm_frac": sig(3 * row_bytes * shard / (mm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)},
        "config5_mul_ctpt_u32": ctpt_block(pk, shard, 32, t_mul, total, me_ms, mac_mul, pmc),
        "host_issue_ms_per_step": round(t_issue_add * 1e3, 4),
        "host_issue_note": "time the calling thread spends queueing one CT+CT step over the %d pool entries; %.1f %% of a step"
                           % (N, 100 * t_issue_add / t_add),
        "end_to_end": {"what": "CT+CT through pgpu_modmul on caller-owned host arrays (plain operands: two products, "
                               "H2D + kernel + D2H pipelined in sub-batches over the worker lanes)",
                       "ms_per_step": round(t_e2e * 1e3, 3), "modmuls_per_s": round(total / t_e2e, 1),
                       "host_GBs": round(3 * W * 8 * total / t_e2e / 1e9, 2)},
    }


# ----------------------------------------------------------------------------------------------------------------
# under torch.distributed.run: one process per GPU, RCCL only for the key broadcast
# ----------------------------------------------------------------------------------------------------------------
def run_ranks(args, world):
    import torch
    import torch.distributed as dist
    if args.config != 2:
        raise SystemExit("--config 4|5 run through the in-process pool: invoke bench.py plainly (no torchrun)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # validation-only knobs (exercise this path on a 1-GPU box): BENCH_SINGLE_DEVICE=1 puts every rank on
    # device 0, BENCH_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU)
    if os.environ.get("BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)

    import pailliercryptolib_amd as pa
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    pa.initialize(local_rank)
    L = _capi.lib()
    nw, pw = KEY_BITS // 64, KEY_BITS // 128

    # ---- key material: rank 0 owns it, everyone else receives it over RCCL ----
    key_words = torch.zeros(2 * pw + 2 * nw, dtype=torch.int64, device="cuda")   # p | q | hs
    if rank == 0:
        p, q, hs = iso_key()
        flat = np.concatenate([ints_to_limbs([p], pw)[0], ints_to_limbs([q], pw)[0], ints_to_limbs([hs], 2 * nw)[0]])
        key_words.copy_(torch.from_numpy(flat.view(np.int64)))
    dist.broadcast(key_words, src=0)
    kw = key_words.cpu().numpy().view(np.uint64)
    p, q, hs = limbs_to_ints(kw[:pw])[0], limbs_to_ints(kw[pw:2 * pw])[0], limbs_to_ints(kw[2 * pw:])[0]
    n = p * q
    pk, sk = pa.PublicKey(n, KEY_BITS, hs=hs), pa.PrivateKey(p, q)

    # the same step as the pool path (run_pool): resident batches of the rank's own GPU (a one-entry pool), ciphertexts
    # stay in the library's device-side form, two batches in flight on its two batch lanes
    m_host, r_host = synth(rank, BATCH, nw, pw)
    B = Batches(L, _capi.check)
    nfl = args.in_flight
    sets = []
    for ln in range(nfl):
        _capi.check(L.pgpu_set_batch_lane(ln))
        sets.append((B.up(m_host), B.up(r_host)))
    _capi.check(L.pgpu_set_batch_lane(0))
    state = {"c": [None] * nfl, "out": [None] * nfl, "i": 0}

    def step():
        k = state["i"] % nfl
        state["i"] += 1
        B.free(state["c"][k], state["out"][k])
        state["c"][k] = B.op(L.pgpu_batch_encrypt, pk._h, sets[k][0], sets[k][1], 64 * pw)
        state["out"][k] = B.op(L.pgpu_batch_decrypt_crt, sk._h, state["c"][k])

    def sync_all():
        _capi.check(L.pgpu_synchronize())
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(2 * nfl if nfl > 1 else 0):      # priming, as in run_pool: every lane reaches its steady-state forms
        step()
    sync_all()
    for _ in range(max(args.warmup, nfl)):
        step()
    sync_all()
    _capi.check(L.pgpu_set_timing(1 if nfl == 1 else 0))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if nfl == 1:
        per_kind = collect_timing(L, 4 * args.steps + 8)
    else:                       # per-kernel times from a short pass on ONE lane (launches do not overlap in it)
        time.sleep(0.15)        # (the other lanes count as active for 50 ms after they last worked: let that lapse)
        for _ in range(2):
            state["i"] = 0
            step()
        _capi.check(L.pgpu_synchronize())
        _capi.check(L.pgpu_set_timing(1))
        for _ in range(6):
            state["i"] = 0
            step()
        _capi.check(L.pgpu_synchronize())
        per_kind = collect_timing(L, 64)
    _capi.check(L.pgpu_set_timing(0))
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ok = all(bool(np.array_equal(B.down(o), m_host)) for o in state["out"])
    if rank == 0:
        from oracle import paillier_oracle as orc
        opk = orc.PublicKey(n, KEY_BITS)
        opk.set_djn(hs)
        ok = ok and limbs_to_ints(B.down(state["c"][0])[:3]) == \
            opk.encrypt(limbs_to_ints(m_host[:3]), limbs_to_ints(r_host[:3]))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)                 # every rank's round trip
    ok = bool(flag.item())
    row_limbs = L.pgpu_batch_row_limbs(state["c"][0])
    if not ok:
        raise SystemExit("bench: GPU results differ from the oracle / round trip failed")
    if rank == 0:
        result = headline(args, world, elapsed, per_kind, nw, pw,
                          f"one process per GPU x{world} (torch.distributed {backend}; key broadcast only)",
                          decrypt_kernel(sk, BATCH, nw, KEY_BITS),
                      encrypt_kernel(pk, BATCH, nw, KEY_BITS, int(os.environ.get("PGPU_FB_WINDOW", "13"))))
        result["config"]["secret_exponent_policy"] = ["fixed-window", "sliding"][L.pgpu_get_secret_exponent_policy()]
        result["config"]["batches_in_flight_per_gpu"] = nfl
        result["config"]["secret_table_access"] = "masked" if L.pgpu_get_table_gather_policy() else "indexed"
        result["config"]["resident_ciphertext_form"] = ("pair rows (%d limbs)" % row_limbs) if row_limbs else "Montgomery-form words"
        if nfl == 2:
            result["config"]["workload"] += ("; %d batches in flight per GPU: consecutive steps rotate over %d of the "
                                             "library's batch lanes (streams), exactly K steps timed" % (nfl, nfl))
            result["config"]["lane_priming_steps"] = 2 * nfl
        bench_line.emit(result)
    B.free(*state["c"], *state["out"], *[h for pair in sets for h in pair])
    dist.barrier()
    dist.destroy_process_group()
    pa.terminate()


def geo_name(in_words, mod_bits, count):
    """modexp / fb_encrypt kernel instantiation the library launches for this shape (pgpu_kernel_geometry)."""
    from pailliercryptolib_amd import _capi
    g, k = ctypes.c_int(), ctypes.c_int()
    _capi.check(_capi.lib().pgpu_kernel_geometry(in_words, mod_bits, count, ctypes.byref(g), ctypes.byref(k)))
    return f"Geo<{g.value},{k.value}>"


def cpu_baseline(n, p, q, hs, m_host, r_host):
    """The oracle's CPU restatements (kind "port") timed on this box's host cores on a bounded sample of the
    same workload (same key, first S elements of the same batch, encrypt + CRT decrypt, results checked).
    Three legs, each the full reference flow (modexps + the host glue of pub_key.cpp:88-105 /
    pri_key.cpp:128-157), OpenMP over the batch like ippMBModExpWrapper (mod_exp.cpp:597-636):
      ifma    oracle/ifma_oracle.c   8 lanes per zmm, radix 2^52, vpmadd52 -- the kind of code the reference's
                                     mbx_exp_mb8 path runs (a restatement, not IPP-Crypto); needs avx512ifma
      openssl BN_mod_exp_mont        the oracle of the reference's own QAT tests (BASELINE.md "B2")
      scalar  oracle/modexp_oracle.c 64-bit CIOS Montgomery + 5-bit fixed window
    Every leg is timed with all usable cores and with ONE thread.  "value" is the fastest all-core leg."""
    from oracle import c_oracle
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd.limbs import ints_to_limbs
    nw, pw = m_host.shape[1], r_host.shape[1]
    sk = orc.PrivateKey(n, p, q)
    threads = min(c_oracle.lib().orc_max_threads(), c_oracle.usable_cpus())   # honours the cgroup CPU quota
    args = [ints_to_limbs([v], pw)[0] for v in (sk.p, sk.q, sk.hp, sk.hq, sk.pinv)]
    n_l, hs_l = ints_to_limbs([n], nw)[0], ints_to_limbs([hs], 2 * nw)[0]

    def flows(backend):
        if backend is None:
            return (lambda m, r: c_oracle.paillier_encrypt(n_l, hs_l, m, r),
                    lambda c: c_oracle.paillier_decrypt_crt(*args, c))
        return (lambda m, r: c_oracle.paillier_encrypt_with(backend, n_l, hs_l, m, r),
                lambda c: c_oracle.paillier_decrypt_crt_with(backend, *args, c))

    def measure(backend, target_s, nthreads):
        enc, dec = flows(backend)
        c_oracle.set_threads(nthreads)

        def run(S):
            t0 = time.perf_counter()
            m = dec(enc(m_host[:S], r_host[:S]))
            dt = time.perf_counter() - t0
            assert np.array_equal(m, m_host[:S])
            return dt
        probe = max(8 * nthreads, 16)
        dt = run(probe)
        S = int(min(m_host.shape[0], max(probe, probe * target_s / dt)))
        S -= S % (8 * nthreads) if S > 8 * nthreads else 0
        dt = run(S)
        reps = 1
        if S == m_host.shape[0] and dt < 0.6 * target_s:      # whole batch too short: repeat it
            reps = int(min(16, max(1, round(target_s / dt))))
            dt = sum(run(S) for _ in range(reps))
        return {"value": round(3 * S * reps / dt, 1), "elements": S * reps, "seconds": round(dt, 2)}

    legs = {}
    backends = []
    fb_info = None
    if c_oracle.ifma_lib() is not None and hasattr(c_oracle.ifma_lib(), "orc_ifma_fb_build"):
        # like for like with the GPU step: hs^r as a fixed-base product over a per-key table (built here, outside the timed
        # samples, as the GPU's is), the decrypt leg by square-and-multiply
        t0 = time.perf_counter()
        nsq_l = ints_to_limbs([n * n], 2 * nw)[0]
        fbw = int(os.environ.get("BENCH_CPU_FB_WINDOW", "8"))
        fb = c_oracle.IfmaFixedBase(hs_l, nsq_l, 64 * pw, fbw)
        fb_info = {"window": fbw, "table_bytes": ((64 * pw + fbw - 1) // fbw) * (1 << fbw) * ((2 * KEY_BITS + 2 + 51) // 52) * 8,
                   "build_s": round(time.perf_counter() - t0, 3)}

        def fb_backend(base, exp, mod):
            return fb(base, exp) if base.shape[1] == 2 * nw and exp.shape[1] == pw else c_oracle.ifma_modexp_batch(base, exp, mod)
        backends.append(("ifma_fixed_base", fb_backend, 6.0))
    if c_oracle.ifma_lib() is not None:
        backends.append(("ifma", c_oracle.ifma_modexp_batch, 6.0))
    if c_oracle.openssl_lib() is not None:
        backends.append(("openssl", c_oracle.openssl_modexp_batch, 4.0))
    backends.append(("scalar", None, 5.0))
    for name, be, target in backends:
        legs[name] = measure(be, target, threads)
        legs[name]["one_thread"] = measure(be, 2.0, 1)
    c_oracle.set_threads(threads)
    best = max(legs, key=lambda k: legs[k]["value"])
    if fb_info:
        legs["ifma_fixed_base"]["fixed_base_table"] = fb_info
    what = {"ifma_fixed_base": "oracle/ifma_oracle.c with hs^r as a fixed-base product (orc_ifma_fb_*: the GPU step's algorithm on "
                               "the CPU; decrypt by square-and-multiply)",
            "ifma": "oracle/ifma_oracle.c (8-lane AVX512-IFMA radix-2^52 restatement of the reference's mb8 path)",
            "openssl": "OpenSSL BN_mod_exp_mont", "scalar": "oracle/modexp_oracle.c (64-bit CIOS)"}[best]
    return {"value": legs[best]["value"], "unit": "modexps/s", "cores": threads, "kind": "port",
            "encrypt_like_for_like": best == "ifma_fixed_base",
            "encrypt_note": "legs.ifma_fixed_base computes hs^r as the GPU step does -- a fixed-base product over a per-key table "
                            "built outside the timed region (CPU: w = 8, 127 products; GPU: w = 13, 78 products); the other legs "
                            "square-and-multiply it (1259 products per element) as the reference's ippMBModExp does",
            "sample": f"{legs[best]['elements']} elements (the same batch, from its start, repeated if shorter than the "
                      f"time target), encrypt + CRT decrypt "
                      f"({3 * legs[best]['elements']} modexps) in {legs[best]['seconds']} s; {what}, gcc -O3 "
                      f"-fopenmp; {threads} OpenMP threads = min(affinity {len(os.sched_getaffinity(0))}, "
                      f"cgroup cpu quota {c_oracle.usable_cpus()}); every leg also timed with 1 thread (legs.*.one_thread)",
            "leg": best, "one_thread_value": legs[best]["one_thread"]["value"], "legs": legs}


if __name__ == "__main__":
    main()
