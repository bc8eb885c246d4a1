"""Kernel timeline of a short burst of resident steps on k batch lanes, from the library's own HIP events
(pgpu_timing_collect_trace).  usage: python tools/probe_trace.py [lanes] [steps] [masked gather 0/1]   (tools/, diagnostics only)"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
gather = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pa.initialize(0)
L = _capi.lib()
if gather:
    _capi.check(L.pgpu_set_table_gather_policy(1))
k = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "iso_kat.json")))
p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
n = p * q
pk, sk = pa.PublicKey(n, 2048, hs=hs), pa.PrivateKey(p, q)
rng = np.random.default_rng(1)
m = np.frombuffer(rng.bytes(8192 * 256), dtype=np.uint64).reshape(8192, 32).copy()
m[:, -1] &= np.uint64((1 << 62) - 1)
r = np.frombuffer(rng.bytes(8192 * 128), dtype=np.uint64).reshape(8192, 16).copy()
ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def up(a):
    h = ctypes.c_void_p()
    _capi.check(L.pgpu_batch_upload(ptr(a), a.shape[0], a.shape[1], a.shape[1], ctypes.byref(h)))
    return h


def op(fn, *a):
    h = ctypes.c_void_p()
    _capi.check(fn(*a, ctypes.byref(h)))
    return h


sets = []
for ln in range(nl):
    _capi.check(L.pgpu_set_batch_lane(ln))
    sets.append((up(m), up(r)))
_capi.check(L.pgpu_set_batch_lane(0))
st = {"c": [None] * nl, "o": [None] * nl, "i": 0}


def step():
    kk = st["i"] % nl
    st["i"] += 1
    for h in (st["c"][kk], st["o"][kk]):
        if h:
            L.pgpu_batch_destroy(h)
    st["c"][kk] = op(L.pgpu_batch_encrypt, pk._h, sets[kk][0], sets[kk][1], 1024)
    st["o"][kk] = op(L.pgpu_batch_decrypt_crt, sk._h, st["c"][kk])


for _ in range(max(3, nl)):
    step()
_capi.check(L.pgpu_synchronize())
_capi.check(L.pgpu_set_timing(1))
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_issue = time.perf_counter() - t0
_capi.check(L.pgpu_synchronize())
dt = time.perf_counter() - t0
cap = 8 * steps + 16
kinds, forms, lanes = (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
start, dur = (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
got = L.pgpu_timing_collect_trace(kinds, forms, lanes, start, dur, cap)
print(f"lanes {nl} steps {steps}: {dt / steps * 1e3:.3f} ms/step, host issue {t_issue / steps * 1e6:.0f} us/step")
names = {1: "dec", 3: "crt", 4: "enc"}
for i in range(got):
    print(f"  lane {lanes[i]} {names.get(kinds[i], kinds[i]):3s} form {forms[i]:2d} start {start[i]:8.3f} end {start[i] + dur[i]:8.3f} dur {dur[i]:7.3f}")
pa.terminate()
