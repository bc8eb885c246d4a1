"""CRT-decrypt kernel time over batch sizes (resident ciphertexts; the library's own HIP-event timers).
usage: python tools/bench_decrypt_sizes.py [--bits 2048|3072] [--ps 0|1|2] [count ...]   (tools/, diagnostics only)
--ps: the policy of the one-lane product-scanning form (pgpu_debug_set_ps_decrypt: 0 never, 1 by size (default), 2 always)"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
pa.initialize(0)
L = _capi.lib()
args = sys.argv[1:]
BITS, PS = 2048, None
while args and args[0].startswith("--"):
    if args[0] == "--bits":
        BITS = int(args[1])
    elif args[0] == "--ps":
        PS = int(args[1])
    args = args[2:]
GOLD = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
if BITS == 2048:
    k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
    p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
else:
    k = [c for c in json.load(open(os.path.join(GOLD, "seeded_vectors.json")))["cases"] if c["bits"] == BITS and c["djn"]][0]
    p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["hs"], 16)
n = p * q
NW = BITS // 64
pk, sk = pa.PublicKey(n, BITS, hs=hs), pa.PrivateKey(p, q)
if PS is not None:
    L.pgpu_debug_set_ps_decrypt(PS)
sizes = [int(a) for a in args] or [8192, 16384, 32768, 65536]
print("key bits", BITS, "ps policy", PS, flush=True)
rng = np.random.default_rng(1)


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


for count in sizes:
    m = np.zeros((count, NW), dtype=np.uint64)
    m[:, 0] = rng.integers(0, 1 << 62, size=count, dtype=np.uint64)
    r = np.frombuffer(rng.bytes(count * NW * 4), dtype=np.uint64).reshape(count, NW // 2).copy()
    hm, hr, c, d = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    _capi.check(L.pgpu_batch_upload(ptr(m), count, NW, NW, ctypes.byref(hm)))
    _capi.check(L.pgpu_batch_upload(ptr(r), count, NW // 2, NW // 2, ctypes.byref(hr)))
    _capi.check(L.pgpu_batch_encrypt(pk._h, hm, hr, BITS // 2, ctypes.byref(c)))
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
    out = np.empty((count, NW), dtype=np.uint64)
    _capi.check(L.pgpu_batch_download(outs[-1], ptr(out)))
    assert np.array_equal(out, m)
    print(count, "decrypt kernel ms", [round(x, 3) for x in dec], "per 8192:", round(min(dec) * 8192 / count, 3), flush=True)
    for h in [hm, hr, c] + outs:
        L.pgpu_batch_destroy(h)
pa.terminate()
