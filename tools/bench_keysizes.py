"""Resident encrypt / CT x PT / decrypt per key class, split form on and off (library HIP-event timers summed per
operation; tools/, diagnostics only).  usage: python tools/bench_keysizes.py [count]"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
pa.initialize(0)
L = _capi.lib()
L.pgpu_debug_set_hensel.argtypes = [ctypes.c_int]
count = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
cases = {c["bits"]: c for c in json.load(open(os.path.join(G, "seeded_vectors.json")))["cases"] if c["djn"]}
rng = np.random.default_rng(2)
_k4 = json.load(open(os.path.join(G, "primes_4096.json")))
_n4 = int(_k4["p"], 16) * int(_k4["q"], 16)
cases[4096] = {"p": _k4["p"], "q": _k4["q"], "hs": hex(pow(_n4 * _n4 - 9, _n4, _n4 * _n4))}   # hs = (-x^2)^n, x = 3


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def timed(fn):
    fn()
    _capi.check(L.pgpu_synchronize())
    t0 = time.perf_counter()
    h = fn()
    _capi.check(L.pgpu_synchronize())
    return (time.perf_counter() - t0) * 1e3, h


for bits in (1024, 2048, 3072, 4096):
    c = cases[bits]
    p, q, hs = int(c["p"], 16), int(c["q"], 16), int(c["hs"], 16)
    n = p * q
    nw = bits // 64
    m = np.zeros((count, nw), dtype=np.uint64)
    m[:, 0] = rng.integers(0, 1 << 62, size=count, dtype=np.uint64)
    r = np.frombuffer(rng.bytes(count * nw * 4), dtype=np.uint64).reshape(count, nw // 2).copy()
    e = rng.integers(0, 1 << 32, size=(count, 1), dtype=np.uint64)
    for mode in (0, 1):
        L.pgpu_debug_set_hensel(mode)
        pk, sk = pa.PublicKey(n, bits, hs=hs), pa.PrivateKey(p, q)
        hm, hr, he = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _capi.check(L.pgpu_batch_upload(ptr(m), count, nw, nw, ctypes.byref(hm)))
        _capi.check(L.pgpu_batch_upload(ptr(r), count, nw // 2, nw // 2, ctypes.byref(hr)))
        _capi.check(L.pgpu_batch_upload(ptr(e), count, 1, 1, ctypes.byref(he)))

        def enc():
            h = ctypes.c_void_p()
            _capi.check(L.pgpu_batch_encrypt(pk._h, hm, hr, bits // 2, ctypes.byref(h)))
            return h
        t_enc, ct = timed(enc)

        def mul():
            h = ctypes.c_void_p()
            _capi.check(L.pgpu_batch_ct_mul(pk._h, ct, he, 32, ctypes.byref(h)))
            return h
        t_mul, cm = timed(mul)

        def add():
            h = ctypes.c_void_p()
            _capi.check(L.pgpu_batch_ct_add(pk._h, ct, cm, ctypes.byref(h)))
            return h
        t_add, _ = timed(add)

        def dec():
            h = ctypes.c_void_p()
            _capi.check(L.pgpu_batch_decrypt_crt(sk._h, ct, ctypes.byref(h)))
            return h
        t_dec, dm = timed(dec)
        out = np.empty((count, nw), dtype=np.uint64)
        _capi.check(L.pgpu_batch_download(dm, ptr(out)))
        assert np.array_equal(out, m)
        # decrypt leg as a fraction of the int-ALU peak (bench.py's counts: executed by the kernel form that ran, and the
        # useful count of the split form); wall time of the call incl. launch overhead and the CRT kernel
        import bench as _b
        if bits in (1024, 2048, 3072) or (bits == 4096 and mode == 0):
            name, per_exp = _b.decrypt_kernel(sk, count, nw, bits)
            fe = per_exp * 2 * count / (t_dec * 1e-3) / 1e12 / _b.PEAK_TMAC32
            fu = _b.decrypt_useful_mac32(nw, bits) * 2 * count / (t_dec * 1e-3) / 1e12 / _b.PEAK_TMAC32
            frac = f"   [{name}: {fe:.2f} executed" + (f", {fu:.2f} useful]" if mode else "]")
        else:
            frac = ""
        print(f"{bits}-bit key, {count} elements, split form {'on ' if mode else 'off'}: encrypt {t_enc:8.2f} ms   "
              f"CT x PT (u32) {t_mul:8.2f} ms   CT + CT {t_add:7.3f} ms   decrypt {t_dec:8.2f} ms{frac}", flush=True)
        del pk, sk
pa.terminate()
