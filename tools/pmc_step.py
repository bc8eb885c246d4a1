"""The resident headline step on k batch lanes for a few rounds -- the command bench.py runs under `rocprofv3 --pmc` (a pass
of its own per counter, no trace domain) to measure the HBM traffic of the kernels of its timed region in the same run.
usage: python tools/pmc_step.py [lanes] [steps]      (tools/; no torch: a short-lived process)"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pa.initialize(0)
L = _capi.lib()
k = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "iso_kat.json")))
p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
pk, sk = pa.PublicKey(p * q, 2048, hs=hs), pa.PrivateKey(p, q)
rng = np.random.default_rng(1)
m = np.frombuffer(rng.bytes(8192 * 256), dtype=np.uint64).reshape(8192, 32).copy()
m[:, -1] &= np.uint64((1 << 62) - 1)
r = np.frombuffer(rng.bytes(8192 * 128), dtype=np.uint64).reshape(8192, 16).copy()


def up(a):
    h = ctypes.c_void_p()
    _capi.check(L.pgpu_batch_upload(a.ctypes.data_as(ctypes.c_void_p), a.shape[0], a.shape[1], a.shape[1], ctypes.byref(h)))
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
held = [[None, None] for _ in range(nl)]
for i in range(2 * nl + steps):
    kk = i % nl
    for h in held[kk]:
        if h:
            L.pgpu_batch_destroy(h)
    c = op(L.pgpu_batch_encrypt, pk._h, sets[kk][0], sets[kk][1], 1024)
    held[kk] = [c, op(L.pgpu_batch_decrypt_crt, sk._h, c)]
_capi.check(L.pgpu_synchronize())
pa.terminate()
