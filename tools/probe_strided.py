"""pgpu_batch_download vs pgpu_batch_download_strided into a pinned block, for an encrypt result (pair rows) and a decrypt
result (plain words), 8192 x 2048-bit.  (tools/, diagnostics only)"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
pa.initialize(0)
L = _capi.lib()
k = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "iso_kat.json")))
p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
pk, sk = pa.PublicKey(p * q, 2048, hs=hs), pa.PrivateKey(p, q)
B = 8192
rng = np.random.default_rng(1)
m = np.frombuffer(rng.bytes(B * 256), dtype=np.uint64).reshape(B, 32).copy()
m[:, -1] &= np.uint64((1 << 62) - 1)
r = np.frombuffer(rng.bytes(B * 128), dtype=np.uint64).reshape(B, 16).copy()
ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def op(fn, *a):
    h = ctypes.c_void_p()
    _capi.check(fn(*a, ctypes.byref(h)))
    return h


bm, br = op(L.pgpu_batch_upload, ptr(m), B, 32, 32), op(L.pgpu_batch_upload, ptr(r), B, 16, 16)
c = op(L.pgpu_batch_encrypt, pk._h, bm, br, 1024)
d = op(L.pgpu_batch_decrypt_crt, sk._h, c)
_capi.check(L.pgpu_synchronize())
pp = ctypes.c_void_p()
_capi.check(L.pgpu_host_alloc(8 << 20, ctypes.byref(pp)))
for name, h, words in (("encrypt result (pair rows)", c, 64), ("decrypt result (words)", d, 32)):
    for label, fn in (("contiguous", lambda: L.pgpu_batch_download(h, pp)), ("strided +2", lambda: L.pgpu_batch_download_strided(h, pp, words + 2))):
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter()
            _capi.check(fn())
            best = min(best, time.perf_counter() - t0)
        print(f"{name:28s} {label:12s} {best * 1e6:8.1f} us")
