"""Probe: CRT decrypt of a few random ciphertexts per key class through the full-width path and both split forms,
each against the oracle (tools/, diagnostics only)."""
import ctypes, json, os, random, sys
sys.path.insert(0, ".")
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import _capi
from oracle import paillier_oracle as orc
pa.initialize()
L = _capi.lib()
L.pgpu_debug_set_hensel.argtypes = [ctypes.c_int]
G = "tests/golden"
keys = {}
for c in json.load(open(f"{G}/seeded_vectors.json"))["cases"]:
    keys[c["bits"]] = (int(c["p"], 16), int(c["q"], 16))
k4 = json.load(open(f"{G}/primes_4096.json"))
keys[4096] = (int(k4["p"], 16), int(k4["q"], 16))
rng = random.Random(1)
for bits, (p, q) in sorted(keys.items()):
    n = p * q
    sk = pa.PrivateKey(p, q)
    osk = orc.PrivateKey(n, p, q)
    for count in (3, 2100):
        c = [rng.randrange(1, n * n) for _ in range(count)]
        want = osk.decrypt(c[:3]) + osk.decrypt(c[-1:])
        for mode in (0, 2, 3):
            L.pgpu_debug_set_hensel(mode)
            got = sk.decrypt(c)
            print(bits, count, "mode", mode, "ok" if got[:3] + got[-1:] == want else "WRONG", flush=True)
pa.terminate()
