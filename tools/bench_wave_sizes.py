"""CRT decrypt of small resident batches: the wavefront-wide latency form (csrc/hensel_wave.hpp) forced on / off, wall time of
a synchronised call, best of 5.  usage: bench_wave_sizes.py [bits=2048] [counts...]   (tools/, diagnostics only)"""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
counts = [int(v) for v in sys.argv[2:]] or [16, 256, 512, 640, 768, 1024, 1536, 2048]
pa.initialize(0)
L = _capi.lib()
GOLD = os.path.join(ROOT, "tests", "golden")
if bits == 2048:
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
else:
    k = [c for c in json.load(open(os.path.join(GOLD, "seeded_vectors.json")))["cases"] if c["bits"] == bits and c["djn"]][0]
    p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["hs"], 16)
pk, sk = pa.PublicKey(p * q, bits, hs=hs), pa.PrivateKey(p, q)
nw = bits // 64
ptr = lambda x: x.ctypes.data_as(ctypes.c_void_p)
def op(fn, *a):
    h = ctypes.c_void_p(); _capi.check(fn(*a, ctypes.byref(h))); return h
rng = np.random.default_rng(1)
for count in counts:
    m = np.frombuffer(rng.bytes(count * nw * 8), dtype=np.uint64).reshape(count, nw).copy(); m[:, -1] &= np.uint64((1 << 62) - 1)
    r = np.frombuffer(rng.bytes(count * nw * 4), dtype=np.uint64).reshape(count, nw // 2).copy()
    c = op(L.pgpu_batch_encrypt, pk._h, op(L.pgpu_batch_upload, ptr(m), count, nw, nw), op(L.pgpu_batch_upload, ptr(r), count, nw // 2, nw // 2), bits // 2)
    res = {}
    for pol in (0, 2):
        L.pgpu_debug_set_wave_decrypt(pol)
        best = 1e9
        for _ in range(6):
            _capi.check(L.pgpu_synchronize()); t0 = time.perf_counter()
            d = op(L.pgpu_batch_decrypt_crt, sk._h, c); _capi.check(L.pgpu_synchronize())
            best = min(best, time.perf_counter() - t0); L.pgpu_batch_destroy(d)
        res[pol] = best * 1e3
    L.pgpu_debug_set_wave_decrypt(1)
    print(f"{bits}-bit key, {count:5d} ciphertexts: multi-lane forms {res[0]:7.3f} ms   wavefront-wide form {res[2]:7.3f} ms", flush=True)
pa.terminate()
