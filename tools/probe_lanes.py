"""Resident steps with 1..4 batches in flight on the library's batch lanes (diagnostics; tools/).
usage: python tools/probe_lanes.py [--policy P] [--steps K] [--count N]
Per number of lanes: ms per step of (a) decrypt-only steps and (b) encrypt + decrypt steps, and which kernel forms ran
(pgpu_timing_collect_ex).  PGPU_SEQ_DECRYPT / --policy: 0 paired kernels only, 1 by launch size, 3 round-3 two-lane mode,
4 adaptive (default)."""
import argparse, collections, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--policy", type=int, default=None)
ap.add_argument("--steps", type=int, default=24)
ap.add_argument("--count", type=int, default=8192)
ap.add_argument("--lanes", type=int, nargs="*", default=[1, 2, 3, 4])
ap.add_argument("--gather", type=int, default=0, help="pgpu_set_table_gather_policy")
ap.add_argument("--bits", type=int, default=2048, help="key size: 2048 (the ISO key), 1024 / 3072 (the seeded DJN fixtures)")
ap.add_argument("--ps", type=int, default=None, help="PGPU_PS_DECRYPT policy (hensel_ps.hpp): 0 never, 1 adaptive, 2 always")
args = ap.parse_args()
pa.initialize(0)
L = _capi.lib()
if args.policy is not None:
    L.pgpu_debug_set_seq_decrypt(args.policy)
if args.gather:
    _capi.check(L.pgpu_set_table_gather_policy(1))
if args.ps is not None:
    L.pgpu_debug_set_ps_decrypt(args.ps)
print("ps policy", args.ps)
GOLD = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
BITS = args.bits
if BITS == 2048:
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
else:
    k = [c for c in json.load(open(os.path.join(GOLD, "seeded_vectors.json")))["cases"] if c["bits"] == BITS and c["djn"]][0]
    p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["hs"], 16)
NW = BITS // 64
n = p * q
pk, sk = pa.PublicKey(n, BITS, hs=hs), pa.PrivateKey(p, q)
count = args.count
rng = np.random.default_rng(1)
m = np.frombuffer(rng.bytes(count * NW * 8), dtype=np.uint64).reshape(count, NW).copy()
m[:, -1] &= np.uint64((1 << 62) - 1)
r = np.frombuffer(rng.bytes(count * NW * 4), dtype=np.uint64).reshape(count, NW // 2).copy()


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def up(a):
    h = ctypes.c_void_p()
    _capi.check(L.pgpu_batch_upload(ptr(a), a.shape[0], a.shape[1], a.shape[1], ctypes.byref(h)))
    return h


def op(fn, *a):
    h = ctypes.c_void_p()
    _capi.check(fn(*a, ctypes.byref(h)))
    return h


def forms(cap=4096):
    kinds, fm, ms = (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_double * cap)()
    got = L.pgpu_timing_collect_ex(kinds, fm, ms, cap)
    out = collections.defaultdict(list)
    for i in range(got):
        out[(kinds[i], fm[i])].append(ms[i])
    return {f"kind{k}/form{f}": (len(v), round(float(np.mean(v)), 3)) for (k, f), v in sorted(out.items())}


print("policy", L.pgpu_debug_get_seq_decrypt(), "count", count, "key bits", BITS, flush=True)
for nl in args.lanes:
    sets = []
    for ln in range(nl):
        _capi.check(L.pgpu_set_batch_lane(ln))
        bm, br = up(m), up(r)
        sets.append((bm, br, op(L.pgpu_batch_encrypt, pk._h, bm, br, BITS // 2)))
    _capi.check(L.pgpu_set_batch_lane(0))
    _capi.check(L.pgpu_synchronize())
    st = {"c": [None] * nl, "o": [None] * nl, "i": 0}

    def free(*hs):
        for h in hs:
            if h:
                L.pgpu_batch_destroy(h)

    def dec_step():
        kk = st["i"] % nl
        st["i"] += 1
        free(st["o"][kk])
        st["o"][kk] = op(L.pgpu_batch_decrypt_crt, sk._h, sets[kk][2])

    def full_step():
        kk = st["i"] % nl
        st["i"] += 1
        free(st["c"][kk], st["o"][kk])
        st["c"][kk] = op(L.pgpu_batch_encrypt, pk._h, sets[kk][0], sets[kk][1], BITS // 2)
        st["o"][kk] = op(L.pgpu_batch_decrypt_crt, sk._h, st["c"][kk])

    for name, fn in (("decrypt-only", dec_step), ("encrypt+decrypt", full_step)):
        for _ in range(2 * nl):
            fn()
        _capi.check(L.pgpu_synchronize())
        _capi.check(L.pgpu_set_timing(1))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        _capi.check(L.pgpu_synchronize())
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        _capi.check(L.pgpu_set_timing(0))
        f = forms()
        out = np.empty((count, NW), dtype=np.uint64)
        for o in st["o"]:
            if o:
                _capi.check(L.pgpu_batch_download(o, ptr(out)))
                assert np.array_equal(out, m), "round trip failed"
        print(f"lanes {nl} {name:16s} {dt:7.3f} ms/step  {3 * count / dt / 1e3 if name != 'decrypt-only' else 2 * count / dt / 1e3:7.3f} M modexps/s  forms {f}", flush=True)
    free(*st["c"], *st["o"], *[h for s3 in sets for h in s3])
pa.terminate()
