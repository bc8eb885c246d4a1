"""Resident DJN encrypt (2048-bit ISO key) at several batch sizes: ms per launch, average over 30 back-to-back launches
(tools/, diagnostics only).  usage: python tools/bench_encrypt_sizes.py 16384 32768 ..."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
pa.initialize(0)
L = _capi.lib()
G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
k = json.load(open(os.path.join(G, "iso_kat.json")))
p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
n = p * q
pk = pa.PublicKey(n, 2048, hs=hs)
rng = np.random.default_rng(3)
ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
for count in [int(a) for a in sys.argv[1:]]:
    m = np.zeros((count, 32), dtype=np.uint64)
    m[:, 0] = rng.integers(0, 1 << 62, size=count, dtype=np.uint64)
    r = np.frombuffer(rng.bytes(count * 128), dtype=np.uint64).reshape(count, 16).copy()
    hm, hr = ctypes.c_void_p(), ctypes.c_void_p()
    _capi.check(L.pgpu_batch_upload(ptr(m), count, 32, 32, ctypes.byref(hm)))
    _capi.check(L.pgpu_batch_upload(ptr(r), count, 16, 16, ctypes.byref(hr)))
    res = []
    for rep in range(3):
        hs_ = []
        for i in range(34):
            if i == 4:
                _capi.check(L.pgpu_synchronize())
                t0 = time.perf_counter()
            h = ctypes.c_void_p()
            _capi.check(L.pgpu_batch_encrypt(pk._h, hm, hr, 1024, ctypes.byref(h)))
            hs_.append(h)
        _capi.check(L.pgpu_synchronize())
        res.append((time.perf_counter() - t0) / 30 * 1e3)
        for h in hs_:
            L.pgpu_batch_destroy(h)
    print(count, "encrypt ms per launch", [round(x, 3) for x in res], flush=True)
pa.terminate()
