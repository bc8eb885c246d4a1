"""CRT-decrypt kernel time over batch sizes (resident ciphertexts; the library's own HIP-event timers).
usage: python tools/bench_decrypt_sizes.py [count ...]   (tools/, diagnostics only)"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
pa.initialize(0)
L = _capi.lib()
k = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "iso_kat.json")))
p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
n = p * q
pk, sk = pa.PublicKey(n, 2048, hs=hs), pa.PrivateKey(p, q)
sizes = [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768, 65536]
rng = np.random.default_rng(1)


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


for count in sizes:
    m = np.zeros((count, 32), dtype=np.uint64)
    m[:, 0] = rng.integers(0, 1 << 62, size=count, dtype=np.uint64)
    r = np.frombuffer(rng.bytes(count * 128), dtype=np.uint64).reshape(count, 16).copy()
    hm, hr, c, d = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    _capi.check(L.pgpu_batch_upload(ptr(m), count, 32, 32, ctypes.byref(hm)))
    _capi.check(L.pgpu_batch_upload(ptr(r), count, 16, 16, ctypes.byref(hr)))
    _capi.check(L.pgpu_batch_encrypt(pk._h, hm, hr, 1024, ctypes.byref(c)))
    _capi.check(L.pgpu_batch_decrypt_crt(sk._h, c, ctypes.byref(d)))      # warm-up
    L.pgpu_batch_destroy(d)
    _capi.check(L.pgpu_synchronize())
    _capi.check(L.pgpu_set_timing(1))
    outs = []
    for _ in range(3):
        d = ctypes.c_void_p()
        _capi.check(L.pgpu_batch_decrypt_crt(sk._h, c, ctypes.byref(d)))
        outs.append(d)
    _capi.check(L.pgpu_synchronize())
    kinds = (ctypes.c_int * 64)()
    ms = (ctypes.c_double * 64)()
    got = L.pgpu_timing_collect(kinds, ms, 64)
    _capi.check(L.pgpu_set_timing(0))
    dec = [ms[i] for i in range(got) if kinds[i] == 1]      # PGPU_KERNEL_MODEXP
    out = np.empty((count, 32), dtype=np.uint64)
    _capi.check(L.pgpu_batch_download(outs[-1], ptr(out)))
    assert np.array_equal(out, m)
    print(count, "decrypt kernel ms", [round(x, 3) for x in dec], "per 8192:", round(min(dec) * 8192 / count, 3), flush=True)
    for h in [hm, hr, c] + outs:
        L.pgpu_batch_destroy(h)
pa.terminate()
